// temporary stubs (replaced by frontier.hip / insert.hip)
#include "fuelmi_internal.h"
extern "C" int fuelmi_map_input_points(fuelmi_map* m, const float*, int, int, const double*) { fuelmi_set_error("not implemented"); return FUELMI_ELIMIT; }
#define NI { fuelmi_set_error("not implemented"); return FUELMI_ELIMIT; }
extern "C" int fuelmi_frontier_create(fuelmi_map*, const fuelmi_frontier_cfg*, fuelmi_frontier**) NI
extern "C" void fuelmi_frontier_destroy(fuelmi_frontier*) {}
extern "C" int fuelmi_frontier_search(fuelmi_frontier*, int*) NI
extern "C" int fuelmi_frontier_commit(fuelmi_frontier*, int) NI
extern "C" int fuelmi_frontier_count(const fuelmi_frontier*, int) NI
extern "C" int fuelmi_frontier_cluster_size(const fuelmi_frontier*, int, int) NI
extern "C" int fuelmi_frontier_cluster_cells(const fuelmi_frontier*, int, int, int*) NI
extern "C" int fuelmi_frontier_cluster_info(const fuelmi_frontier*, int, int, double*) NI
extern "C" int fuelmi_frontier_removed_count(const fuelmi_frontier*) NI
extern "C" int fuelmi_frontier_removed_ids(const fuelmi_frontier*, int*) NI
extern "C" int fuelmi_frontier_get_flags(fuelmi_frontier*, char*) NI
