// probe: where does global_load_lds_dwordx4 (gfx950) put a lane's 16 bytes?  Assumed: LDS base + lane * 16 (k_vhs_noise's tile fill).
//   hipcc --offload-arch=gfx950 -O2 -o tools/probe_lds128.bin tools/probe_lds128.hip && tools/probe_lds128.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const unsigned *g, unsigned *out)
{
    __shared__ __attribute__((aligned(16))) unsigned s[512];
    for (int i = threadIdx.x; i < 512; i += 64) s[i] = 0xdeadbeefu;
    __syncthreads();
    // lane l asks for the 16 bytes at g + 4 * (3 + 5 * l) dwords: unaligned to 16, 4-byte aligned, not contiguous between lanes
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *) (g + 3 + 5 * threadIdx.x),
                                     (void __attribute__((address_space(3))) *) (s + 64), 16, 0, 0);
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64) out[i] = s[i];
}
int main()
{
    unsigned h[1024], o[512], *g, *d;
    for (int i = 0; i < 1024; i++) h[i] = 1000u + i;
    hipMalloc(&g, sizeof h); hipMalloc(&d, sizeof o);
    hipMemcpy(g, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, g, d);
    hipMemcpy(o, d, sizeof o, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++) for (int j = 0; j < 4; j++) if (o[64 + 4 * l + j] != 1000u + 3 + 5 * l + j) bad++;
    for (int i = 0; i < 64; i++) if (o[i] != 0xdeadbeefu) bad++;
    for (int i = 64 + 256; i < 512; i++) if (o[i] != 0xdeadbeefu) bad++;
    printf("global_load_lds_dwordx4: lane l -> LDS base + 16 l: %d mismatches (lane 1 got %u %u %u %u, expected 1008..1011)\n", bad, o[68], o[69], o[70], o[71]);
    return bad != 0;
}
