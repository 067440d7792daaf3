// Probe of ds_read_b64_tr_b16 (gfx950): what does lane L receive when every lane passes its own LDS address?
// build: hipcc --offload-arch=gfx950 -O2 ds_read_tr.hip -o ds_read_tr ; run on the GPU box
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void probe(int* out, int row_stride_elems) {
  __shared__ short lds[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (short)i;  // value = element index
  __syncthreads();
  const int lane = threadIdx.x;
  // each lane points at row (lane % 16), column block (lane / 16) * 4  of a [row][col] tile with the given row stride
  const short* p = lds + (lane & 15) * row_stride_elems + (lane >> 4) * 4;
  v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)p);
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = v[j];
}
int main() {
  int* d; hipMalloc(&d, 64 * 4 * sizeof(int));
  int h[256];
  for (int rs : {64}) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, rs);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("row stride %d: lane -> 4 values as (row,col)\n", rs);
    for (int l = 0; l < 64; ++l) {
      printf("lane %2d:", l);
      for (int j = 0; j < 4; ++j) printf(" (%2d,%2d)", h[l * 4 + j] / rs, h[l * 4 + j] % rs);
      printf("\n");
    }
  }
  return 0;
}
