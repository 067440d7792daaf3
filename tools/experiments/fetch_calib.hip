// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access shapes this library uses (MI355X_MICROARCH.md: "FETCH_SIZE
// reports exactly 1/2 of the bytes of a wide coalesced streaming read ... other access widths are uncalibrated: calibrate on a
// known byte count in your own access pattern").  Every kernel reads each byte of a 1 GiB buffer (4x the Infinity Cache)
// exactly once by LDS-DMA (global_load_lds_dwordx4), in runs of RUN bytes: consecutive lanes take consecutive 16-byte
// chunks of a run, consecutive runs are PITCH bytes apart (a "row" of a tile), rows of one tile first, then the next tile.
//   wide      : RUN = 1024 (one wave instruction = 1 KiB contiguous)          - the conv halo / 1x1 streaming shape
//   run256    : RUN = 256, PITCH = 5120                                        - 128-channel pixel rows of a 16-wide tile
//   run160    : RUN = 160, PITCH = 1280                                        - the fused stem's patch lines (80 bf16 pixels)
//   run64     : RUN = 64,  PITCH = 2560                                        - 32-channel pixels gathered one by one
//   run32     : RUN = 32,  PITCH = 1024                                        - 16-channel pixels (the 16 -> 16 layers)
// Build: hipcc --offload-arch=gfx950 -O3 fetch_calib.hip -o fetch_calib ; run under rocprofv3 --pmc FETCH_SIZE (tools/pmc_fetch_calib.sh)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int RUN, int PITCH>
__global__ __launch_bounds__(256) void read_runs(const char* buf, size_t bytes, unsigned* sink) {
  __shared__ __attribute__((aligned(16))) char lds[4 * 1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int CH = RUN / 16;                 // 16-byte chunks per run
  constexpr size_t TILE = (size_t)PITCH * (PITCH / RUN);  // a tile = PITCH/RUN column blocks of full rows... see below
  // enumerate 16-byte items: item -> (tile, column block cb, row, chunk): address = tile*TILE + row*PITCH + cb*RUN + chunk*16
  // with rows = PITCH / RUN rows per tile, so that a tile is a dense TILE-byte region read in RUN-wide column blocks
  const size_t items = bytes / 16;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t it = (size_t)blockIdx.x * 256 + wave * 64; it < items; it += stride) {
    const size_t i = it + lane;
    const size_t tile = (i * 16) / TILE;
    const size_t in_tile = i - tile * (TILE / 16);
    const size_t chunk = in_tile % CH;
    const size_t rowi = (in_tile / CH) % (PITCH / RUN);
    const size_t cb = in_tile / CH / (PITCH / RUN);
    const char* src = buf + tile * TILE + rowi * PITCH + cb * RUN + chunk * 16;
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds + wave * 1024), 16, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0 && blockIdx.x == 0) sink[0] = *(unsigned*)lds;
}

template <int RUN, int PITCH>
void run(const char* name, const char* buf, size_t bytes, unsigned* sink) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((read_runs<RUN, PITCH>), dim3(256 * 8), dim3(256), 0, 0, buf, bytes, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-8s RUN %4d PITCH %5d bytes %zu  %.3f ms  %.1f GB/s\n", name, RUN, PITCH, bytes, ms, bytes / ms / 1e6);
}

int main() {
  const size_t bytes = (size_t)1 << 30;
  char* buf; unsigned* sink;
  hipMalloc(&buf, bytes + (1 << 20)); hipMalloc(&sink, 64);  // slack: the last, partial tile of a pattern reads past `bytes`
  hipMemset(buf, 1, bytes + (1 << 20));
  hipDeviceSynchronize();
  run<1024, 1024>("wide", buf, bytes, sink);
  run<256, 5120>("run256", buf, bytes, sink);
  run<160, 1280>("run160", buf, bytes, sink);
  run<64, 2560>("run64", buf, bytes, sink);
  run<32, 1024>("run32", buf, bytes, sink);
  hipDeviceSynchronize();
  return 0;
}
