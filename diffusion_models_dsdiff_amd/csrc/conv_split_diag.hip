// conv_split_diag.hip — the diagnostic instantiations of the dominant convolution kernel (clock stamps around prologue /
// k-loop / epilogue, one cost of the loop removed at a time): never on the product path, a translation unit of their own so
// that they compile beside conv_split.hip.  tools/conv_stamps.py, dsd_bench_conv2d_stamps.
#include "conv_split_kernels.inc"

namespace dsd {

void launch_split_diag(const SplitP& p, dim3 grid, hipStream_t s) {
    switch (p.diag) {
        case 0: hipLaunchKernelGGL((conv_split_ad_kernel<5, 3, false, 2, false, 1>), grid, dim3(256), 0, s, p); break;
        case 2: hipLaunchKernelGGL((conv_split_ad_kernel<5, 3, false, 2, false, 3>), grid, dim3(256), 0, s, p); break;
        case 4: hipLaunchKernelGGL((conv_split_ad_kernel<5, 3, false, 2, false, 5>), grid, dim3(256), 0, s, p); break;
        case 8: hipLaunchKernelGGL((conv_split_ad_kernel<5, 3, false, 2, false, 9>), grid, dim3(256), 0, s, p); break;
        case 16: hipLaunchKernelGGL((conv_split_ad_kernel<5, 3, false, 2, false, 17>), grid, dim3(256), 0, s, p); break;
        case 31: hipLaunchKernelGGL((conv_split_ad_kernel<5, 3, false, 2, false, 32>), grid, dim3(256), 0, s, p); break;
        case 32: hipLaunchKernelGGL((conv_split_ad_kernel<5, 3, false, 2, false, 33>), grid, dim3(256), 0, s, p); break;
        case 256: hipLaunchKernelGGL((conv_split_ad_kernel<5, 3, false, 2, false, 257>), grid, dim3(256), 0, s, p); break;
        case 512: hipLaunchKernelGGL((conv_split_ad_kernel<5, 3, false, 2, false, 1, true>), grid, dim3(256), 0, s, p); break;
        default: fail("conv stamps: what-if %d is not instantiated in this build (0, 2, 4, 8, 16, 31, 32, 256, 512 are; 1, 3, 64, 128 were measured in round 2 and their cases removed to keep the build short: add the case back to re-measure)", p.diag);
    }
}

}  // namespace dsd
