"""Drop-in link proof (north_star: "exploration_manager / plan_manage link against the new library
unchanged").  The reference's own callers of the hot path -- fast_exploration_manager.cpp:88-293,
planner_manager.cpp:96-118,124-316, astar2.cpp:97-109, kinodynamic_astar.cpp:172-177, graph_node.cpp:38 -- are
compiled unmodified against fuel_amd/facade/ and linked with libfuelmi_facade.so (tests/dropin/Makefile).
Every symbol of SDFMap / EDTEnvironment / FrontierFinder / BsplineOptimizer they reference must be defined by
the facade library.  CPU only; needs /root/reference (authoring container)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/fuel_planner"
CLASSES = r"fast_planner::(SDFMap|EDTEnvironment|FrontierFinder|BsplineOptimizer)::"


def _nm(args):
    return subprocess.run(["nm"] + args, capture_output=True, text=True, check=True).stdout.splitlines()


@pytest.fixture(scope="module")
def callers():
    if not os.path.isdir(REF):
        pytest.skip("reference checkout not present")
    subprocess.run(["make", "-C", os.path.join(ROOT, "fuel_amd", "facade"), "-s"], check=True)
    p = subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "dropin"), "-s"], capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stderr[-4000:]
    return os.path.join(ROOT, "build", "dropin")


def test_reference_callers_compile_and_link_against_the_facade(callers):
    objs = sorted(f for f in os.listdir(callers) if f.endswith(".o"))
    assert objs == ["astar2.o", "fast_exploration_manager.o", "graph_node.o", "kinodynamic_astar.o",
                    "planner_manager.o"]
    assert os.path.exists(os.path.join(callers, "libfuel_callers.so"))


def test_no_unresolved_hot_path_symbol(callers):
    """every fast_planner::{SDFMap,EDTEnvironment,FrontierFinder,BsplineOptimizer}:: symbol the callers leave
    undefined is exported by libfuelmi_facade.so (mangled names compared: same signatures, not just same names)"""
    need = set()
    for o in os.listdir(callers):
        if not o.endswith(".o"):
            continue
        mangled = [l.split()[-1] for l in _nm(["-u", os.path.join(callers, o)])]
        demangled = [l.split(None, 1)[-1] for l in _nm(["-u", "-C", os.path.join(callers, o)])]
        for m, d in zip(mangled, demangled):
            if re.search(CLASSES, d):
                need.add((m, d))
    assert len(need) >= 30, "the callers reference the hot-path classes: %d symbols found" % len(need)
    have = {l.split()[-1] for l in _nm(["-D", "--defined-only", os.path.join(ROOT, "fuel_amd", "libfuelmi_facade.so")])}
    missing = sorted(d for m, d in need if m not in have)
    assert not missing, "unresolved against libfuelmi_facade.so:\n" + "\n".join(missing)
    # and the linked caller library itself carries them as imports from the facade, not as local definitions
    und = {l.split()[-1] for l in _nm(["-D", "-u", os.path.join(callers, "libfuel_callers.so")])}
    assert {m for m, _ in need} <= und


def test_callers_use_the_inline_getters_of_the_facade_header(callers):
    """the inline getters (getOccupancy / getInflateOccupancy / getDistance / posToIndex ...) are compiled INTO the
    callers from fuel_amd/facade/plan_env/sdf_map.h -- they appear as weak definitions in the callers' objects"""
    weak = set()
    for o in ("astar2.o", "kinodynamic_astar.o", "graph_node.o", "fast_exploration_manager.o"):
        for l in _nm(["-C", os.path.join(callers, o)]):
            parts = l.split(None, 2)
            if len(parts) == 3 and parts[1] in ("W", "w") and "fast_planner::SDFMap::" in parts[2]:
                weak.add(parts[2].split("(")[0])
    for name in ("fast_planner::SDFMap::getInflateOccupancy", "fast_planner::SDFMap::getOccupancy",
                 "fast_planner::SDFMap::posToIndex"):
        assert name in weak, (name, sorted(weak))
