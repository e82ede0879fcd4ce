// Instantiates the wide-band and full-matrix fused-pair DTW kernels for frame dimension 26.
#include "dtw_wide.h"
#include "dtw_full.h"
namespace apd {
template bool launch_wide<26>(const AlignLaunch &, int, int, hipStream_t, hipError_t *);
template bool launch_full<26>(const AlignLaunch &, bool, int, int, hipStream_t, hipError_t *);
}
