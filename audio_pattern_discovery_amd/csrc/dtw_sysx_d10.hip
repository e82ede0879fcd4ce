// Instantiates the strict-mode systolic kernels (unit penalties, the reference's distance arithmetic operation for operation:
// apd_set_distance_mode 2) for frame dimension 10 -- a unit of their own: parallel builds, own scheduler flags (see the Makefile).
#define APD_SYSTOLIC_STRICT_UNIT
#include "dtw_systolic.h"
namespace apd {
template bool launch_systolic_strict<10>(const AlignLaunch &, int, int, hipStream_t);
}
