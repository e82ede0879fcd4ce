// Instantiates the systolic fused-pair DTW kernels for frame dimension 10 (one unit per D and kernel family: parallel builds,
// per-family compiler flags -- see the Makefile).
#include "dtw_systolic.h"
namespace apd {
template bool launch_systolic<10>(const AlignLaunch &, int, int, bool, hipStream_t);
}
