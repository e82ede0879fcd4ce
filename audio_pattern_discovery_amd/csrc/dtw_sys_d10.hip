// Instantiates the systolic fused-pair DTW kernel for frame dimension 10 (one unit per D so they build in parallel).
#include "dtw_systolic.h"
namespace apd {
template bool launch_systolic<10>(const AlignLaunch &, int, int, bool, hipStream_t);
}
