// hip_floor.hip -- the HIP runtime's own start-up in a one-shot process: initialise, allocate, launch ONE trivial kernel
// (a code object of a few hundred bytes), synchronise, exit.  The wall time of this process is the floor of ANY one-shot
// HIP program on the box (tools/time_cli.py compares the `ntsc` CLI against it).
//   hipcc --offload-arch=gfx950 -O2 -o tools/hip_floor.bin tools/hip_floor.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <time.h>
__global__ void k_one(int *p) { p[threadIdx.x] = threadIdx.x; }
static double now_ms() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return 1e3 * t.tv_sec + 1e-6 * t.tv_nsec; }
int main()
{
    double t0 = now_ms(), t1;
    int n = 0, *p = nullptr;
    (void) hipInit(0);
    t1 = now_ms(); printf("hipInit                       %8.2f ms\n", t1 - t0); t0 = t1;
    (void) hipGetDeviceCount(&n); (void) hipSetDevice(0);
    t1 = now_ms(); printf("hipGetDeviceCount+SetDevice   %8.2f ms\n", t1 - t0); t0 = t1;
    (void) hipMalloc(&p, 1 << 20);
    t1 = now_ms(); printf("first hipMalloc               %8.2f ms\n", t1 - t0); t0 = t1;
    hipLaunchKernelGGL(k_one, dim3(1), dim3(64), 0, 0, p);
    (void) hipDeviceSynchronize();
    t1 = now_ms(); printf("first launch + sync           %8.2f ms\n", t1 - t0); t0 = t1;
    hipLaunchKernelGGL(k_one, dim3(1), dim3(64), 0, 0, p);
    (void) hipDeviceSynchronize();
    t1 = now_ms(); printf("second launch + sync          %8.2f ms\n", t1 - t0);
    return 0;
}
