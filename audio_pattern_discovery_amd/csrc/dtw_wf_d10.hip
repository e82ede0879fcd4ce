// Instantiates the wide-band and full-matrix fused-pair DTW kernels for frame dimension 10.
#include "dtw_wide.h"
#include "dtw_full.h"
namespace apd {
template bool launch_wide<10>(const AlignLaunch &, int, int, hipStream_t, hipError_t *);
template bool launch_full<10>(const AlignLaunch &, bool, int, int, hipStream_t, hipError_t *);
}
