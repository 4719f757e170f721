// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 per access width (round-5 review item 8): streaming kernels that
// read or write a KNOWN number of bytes (1 GiB, four times the 256 MiB Infinity Cache) with 2 / 4 / 8 / 16 bytes per lane, through
// plain global loads and through buffer descriptors -- the access shapes of this library's kernels (two-byte twin loads, dword
// rows, 8- and 16-byte rows, 16-byte drains).  tools/fetch_calib.py turns two rocprofv3 --pmc passes over this program into
// profiles/r06_fetch_calibration.txt (known bytes / counter bytes per kernel).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/fetch_calib.hip -o /tmp/fetch_calib && /tmp/fetch_calib
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr size_t BYTES = size_t(1) << 30;

// every byte of a loaded value is used (or the compiler narrows the load)
__device__ __forceinline__ unsigned fold(unsigned short v) { return v; }
__device__ __forceinline__ unsigned fold(unsigned v) { return v; }
__device__ __forceinline__ unsigned fold(uint2 v) { return v.x ^ v.y; }
__device__ __forceinline__ unsigned fold(uint4 v) { return v.x ^ v.y ^ v.z ^ v.w; }

template <typename V>
__global__ __launch_bounds__(256) void read_global(const V* __restrict__ p, size_t n, unsigned* sink) {
  unsigned acc = 0;
  for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    acc += fold(p[i]);
  }
  if (acc == 0x12345678u) *sink = acc;  // (never: keeps the loads alive without a write stream)
}
template <int W>  // bytes per lane: 2, 4, 8, 16
__global__ __launch_bounds__(256) void read_buffer(const void* p, size_t n, unsigned* sink) {
  // 2^31-byte windows through one descriptor each (the vector offset is 32 bits)
  unsigned acc = 0;
  const size_t per = (size_t)gridDim.x * 256;
  for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += per) {
    const size_t byte = i * W;
    const size_t win = byte >> 30;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(static_cast<const char*>(p)) + (win << 30), 0, 1 << 30, 0x00020000);
    const int off = (int)(byte & ((size_t(1) << 30) - 1));
    if constexpr (W == 2) acc += __builtin_amdgcn_raw_buffer_load_b16(rs, off, 0, 0);
    if constexpr (W == 4) acc += __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0);
    if constexpr (W == 8) {
      const auto v = __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 0);
      acc += v[0] ^ v[1];
    }
    if constexpr (W == 16) {
      const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
      acc += v[0] ^ v[1] ^ v[2] ^ v[3];
    }
  }
  if (acc == 0x12345678u) *sink = acc;
}
// rows of 128 lanes-worth read with a stride: every wave reads 64 x W contiguous bytes, then skips as much (half of every
// 2 x 64 x W span): the pattern of a tile whose rows are shorter than a cache line run
template <typename V>
__global__ __launch_bounds__(256) void read_global_half(const V* __restrict__ p, size_t n, unsigned* sink) {
  unsigned acc = 0;
  for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n / 2; i += (size_t)gridDim.x * 256) {
    const size_t j = (i & 63) + ((i >> 6) << 7);
    acc += fold(p[j]);
  }
  if (acc == 0x12345678u) *sink = acc;
}
template <typename V>
__global__ __launch_bounds__(256) void write_global(V* __restrict__ p, size_t n, V v) {
  for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = v;
}

int main() {
  void* buf = nullptr;
  unsigned* sink = nullptr;
  CK(hipMalloc(&buf, BYTES + 4096));
  CK(hipMalloc((void**)&sink, 64));
  CK(hipMemset(buf, 1, BYTES));
  const int grid = 256 * 8;
  for (int rep = 0; rep < 3; ++rep) {
    read_global<unsigned short><<<grid, 256>>>((const unsigned short*)buf, BYTES / 2, sink);
    read_global<unsigned><<<grid, 256>>>((const unsigned*)buf, BYTES / 4, sink);
    read_global<uint2><<<grid, 256>>>((const uint2*)buf, BYTES / 8, sink);
    read_global<uint4><<<grid, 256>>>((const uint4*)buf, BYTES / 16, sink);
    read_buffer<2><<<grid, 256>>>(buf, BYTES / 2, sink);
    read_buffer<4><<<grid, 256>>>(buf, BYTES / 4, sink);
    read_buffer<8><<<grid, 256>>>(buf, BYTES / 8, sink);
    read_buffer<16><<<grid, 256>>>(buf, BYTES / 16, sink);
    read_global_half<unsigned short><<<grid, 256>>>((const unsigned short*)buf, BYTES / 2, sink);  // 64-byte runs
    read_global_half<unsigned><<<grid, 256>>>((const unsigned*)buf, BYTES / 4, sink);              // 256-byte runs
    write_global<unsigned short><<<grid, 256>>>((unsigned short*)buf, BYTES / 2, (unsigned short)1);
    write_global<unsigned><<<grid, 256>>>((unsigned*)buf, BYTES / 4, 1u);
    write_global<uint2><<<grid, 256>>>((uint2*)buf, BYTES / 8, make_uint2(1, 1));
    write_global<uint4><<<grid, 256>>>((uint4*)buf, BYTES / 16, make_uint4(1, 1, 1, 1));
  }
  CK(hipDeviceSynchronize());
  printf("fetch_calib: %zu bytes per launch (reads and writes; the *_half kernels touch half of them)\n", BYTES);
  return 0;
}
