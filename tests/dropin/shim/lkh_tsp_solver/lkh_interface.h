// stand-in (tests/dropin only): the one entry point of the TSP solver package the exploration manager calls
#ifndef DROPIN_LKH_INTERFACE_H_
#define DROPIN_LKH_INTERFACE_H_
int solveTSPLKH(const char* input_file);
#endif
