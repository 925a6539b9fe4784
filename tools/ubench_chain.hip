// tools/ubench_chain.hip -- how long ONE wave takes per instruction of a dependent chain (the burst integrators of
// k_hsync_wave are such a chain: 1200 steps per field).  hipcc --offload-arch=gfx950 -O3 -o tools/ubench_chain.bin tools/ubench_chain.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int KIND>
__global__ void k(int *out, int n, int seed, long long *ticks)
{
    int a = seed + threadIdx.x, b = seed * 3 + 1;
    long long t0 = wall_clock64();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (KIND == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));
            if (KIND == 1) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a) : "v"(b));
            if (KIND == 2) asm volatile("v_med3_i32 %0, %0, %1, 1" : "+v"(a) : "v"(b));
            if (KIND == 3) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(a));
            if (KIND == 4) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(a) : "v"(b));
            if (KIND == 5) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(a) : "v"(b));
            if (KIND == 6) asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %0" : "+v"(a), "+v"(b));   // two independent-ish
            if (KIND == 7) asm volatile("s_add_u32 %0, %0, 3" : "+s"(seed));
        }
    }
    long long t1 = wall_clock64();
    if (threadIdx.x == 0) { out[blockIdx.x] = a + b + seed; ticks[blockIdx.x] = t1 - t0; }
}
int main(int argc, char **argv)
{
    int *out; long long *ticks, h[1024];
    hipMalloc(&out, 4096); hipMalloc(&ticks, 8 * 1024);
    const int n = 20000;
    const char *names[] = { "v_add_u32", "v_and_b32", "v_med3_i32", "v_ashrrev_i32", "v_add_u32_sdwa", "v_mul_i32_i24", "2 x v_add (pair)", "s_add_u32" };
    for (int blocks = 1; blocks <= 1024; blocks *= 1024) {
        for (int kind = 0; kind < 8; kind++) {
            for (int rep = 0; rep < 2; rep++) {
                switch (kind) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), 0, 0, out, n, 5, ticks); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, out, n, 5, ticks); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(64), 0, 0, out, n, 5, ticks); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(64), 0, 0, out, n, 5, ticks); break;
                case 4: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(64), 0, 0, out, n, 5, ticks); break;
                case 5: hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(64), 0, 0, out, n, 5, ticks); break;
                case 6: hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(64), 0, 0, out, n, 5, ticks); break;
                case 7: hipLaunchKernelGGL(k<7>, dim3(blocks), dim3(64), 0, 0, out, n, 5, ticks); break;
                }
                hipDeviceSynchronize();
            }
            hipMemcpy(h, ticks, 8 * blocks, hipMemcpyDeviceToHost);
            const double ns = h[0] * 10.0 / ((double) n * 16 * (kind == 6 ? 2 : 1));
            printf("%4d block(s) of one wave  %-18s %.2f ns per instruction (= %.1f cycles at 2.4 GHz)\n", blocks, names[kind], ns, ns * 2.4);
        }
    }
    return 0;
}
