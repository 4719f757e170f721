// bf16-mode instantiations of the fused ConvNeXt32 forward (convnext_kernel.h), a translation unit of their own because they are
// compiled with -fno-slp-vectorize (stylish_tts_amd/build.py FILE_FLAGS; the header says why).  Called by launch_convnext32.
#include "convnext_kernel.h"

namespace sty {

void launch_convnext32_bf16(const Cnx32Args& a, dim3 grid, int pass, hipStream_t st) {
  if (pass == 1)
    hipLaunchKernelGGL((convnext32_kernel<false, true>), grid, dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((convnext32_kernel<true, true>), grid, dim3(256), 0, st, a);
}

}  // namespace sty
