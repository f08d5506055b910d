// Probe: exact lane/element semantics of ds_read_b64_tr_b16 on gfx950 (needed for the bf16 wgrad path).
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe_tr16.hip -o tools/probe_tr16 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short* out, int stride_bytes) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const char* base = reinterpret_cast<const char*>(lds) + threadIdx.x * stride_bytes;
    v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(base));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)r[j];
}
int main() {
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    unsigned short h[256];
    int strides[3] = {8, 64, 40};
    for (int s = 0; s < 3; ++s) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, strides[s]);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("stride_bytes=%d (lane addr = lane*stride; element index = bytes/2)\n", strides[s]);
        for (int l = 0; l < 64; ++l) {
            printf(" lane%02d:", l);
            for (int j = 0; j < 4; ++j) {
                int e = h[l * 4 + j];
                int src_lane = (e * 2) / strides[s], off = (e * 2 - src_lane * strides[s]) / 2;
                printf(" %5d(L%02d+%d)", e, src_lane, off);
            }
            printf("%s", (l % 2) ? "\n" : "  |");
        }
    }
    return 0;
}
