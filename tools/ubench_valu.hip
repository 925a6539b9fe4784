// Integer VALU issue-rate microbenchmark for gfx950 (MI355X): cycles a wave64 instruction occupies its
// SIMD, per opcode.  8 independent dependency chains per lane, 8 waves per SIMD, inline asm so the
// compiler can neither fold nor re-select the instruction.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITERS 2048
#define DEF(NAME, ASM) \
__global__ void k_##NAME(int *out, int a, int b) { \
    int x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
    int va = a + (threadIdx.x & 1), vb = b; \
    for (int i = 0; i < ITERS; i++) { \
        asm volatile(ASM : "+v"(x0) : "v"(va), "v"(vb)); asm volatile(ASM : "+v"(x1) : "v"(va), "v"(vb)); \
        asm volatile(ASM : "+v"(x2) : "v"(va), "v"(vb)); asm volatile(ASM : "+v"(x3) : "v"(va), "v"(vb)); \
        asm volatile(ASM : "+v"(x4) : "v"(va), "v"(vb)); asm volatile(ASM : "+v"(x5) : "v"(va), "v"(vb)); \
        asm volatile(ASM : "+v"(x6) : "v"(va), "v"(vb)); asm volatile(ASM : "+v"(x7) : "v"(va), "v"(vb)); \
    } \
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7; }
DEF(add_u32, "v_add_u32 %0, %0, %1")
DEF(sub_u32, "v_sub_u32 %0, %0, %1")
DEF(ashr, "v_ashrrev_i32 %0, 1, %0")
DEF(mad_i24, "v_mad_i32_i24 %0, %0, %1, %2")
DEF(mul_i24, "v_mul_i32_i24 %0, %0, %1")
DEF(mad_u24, "v_mad_u32_u24 %0, %0, %1, %2")
DEF(mul_lo, "v_mul_lo_u32 %0, %0, %1")
DEF(add_sdwa, "v_add_u32_sdwa %0, sext(%1), %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD")
DEF(bfe_i32, "v_bfe_i32 %0, %0, 3, 16")
DEF(perm, "v_perm_b32 %0, %0, %1, %2")
DEF(med3, "v_med3_i32 %0, %0, %1, %2")
DEF(add3, "v_add3_u32 %0, %0, %1, %2")
DEF(lshl_add, "v_lshl_add_u32 %0, %0, 2, %1")
DEF(add_lshl, "v_add_lshl_u32 %0, %0, %1, 4")
DEF(lshl_or, "v_lshl_or_b32 %0, %0, 8, %1")
DEF(fma_f32, "v_fma_f32 %0, %0, %1, %2")
DEF(mad_i32_i16, "v_mad_i32_i16 %0, %0, %1, %2")
DEF(dot2_i32_i16, "v_dot2_i32_i16 %0, %0, %1, %2")
DEF(pk_add_i16, "v_pk_add_i16 %0, %0, %1")
DEF(mul_hi_i32, "v_mul_hi_i32 %0, %0, %1")
__global__ void k_mad_i64_i32(int *out, int a, int b) {
    long x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    int va = a + (threadIdx.x & 1), vb = b;
    for (int i = 0; i < ITERS; i++) {
#define M64(x) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(x) : "v"(va), "v"(vb) : "vcc")
        M64(x0); M64(x1); M64(x2); M64(x3); M64(x4); M64(x5); M64(x6); M64(x7);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (int) (x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7);
}
DEF(mov, "v_mov_b32 %0, %1")
DEF(cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
DEF(and_b32, "v_and_b32 %0, %0, %1")
DEF(lshlrev, "v_lshlrev_b32 %0, 1, %0")
DEF(max_i32, "v_max_i32 %0, %0, %1")
DEF(mul_i24_sdwa, "v_mul_i32_i24_sdwa %0, %1, sext(%0) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0")
DEF(mul_f32, "v_mul_f32 %0, %0, %1")
DEF(floor_f32, "v_floor_f32 %0, %0")
DEF(cvt_f32_ubyte1, "v_cvt_f32_ubyte1 %0, %0")
DEF(cvt_f32_ubyte2, "v_cvt_f32_ubyte2 %0, %1")
DEF(cvt_i32_f32, "v_cvt_i32_f32 %0, %0")
DEF(cvt_f32_i32, "v_cvt_f32_i32 %0, %0")
DEF(cvt_pk_u8_f32, "v_cvt_pk_u8_f32 %0, %1, 1, %0")
DEF(med3_f32, "v_med3_f32 %0, %0, %1, %2")
DEF(max_f32, "v_max_f32 %0, %0, %1")
DEF(dot4_i32_i8, "v_dot4_i32_i8 %0, %0, %1, %2")
DEF(fract_f32, "v_fract_f32 %0, %0")
DEF(sub_f32, "v_sub_f32 %0, %0, %1")
DEF(xor_b32, "v_xor_b32 %0, %0, %1")
template <class K> void run(const char *name, K kern, int *d)
{
    hipEvent_t e0, e1; (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
    const int blocks = 256 * 8, threads = 256;    // 8 blocks x 4 waves per CU = 8 waves / SIMD
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, 3, 5);
    (void) hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, 3, 5);
    (void) hipEventRecord(e1); (void) hipEventSynchronize(e1);
    float ms; (void) hipEventElapsedTime(&ms, e0, e1);
    double winst = (double) blocks * (threads / 64) * ITERS * 8.0;
    double per_simd_per_s = winst / 1024.0 / (ms * 1e-3);
    printf("%-16s %.3f ms  %.3f G wave-instr/s/SIMD  -> %.2f cycles per wave64 instr @2.4 GHz\n", name, ms, per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s);
}
#define RUN(NAME) run(#NAME, k_##NAME, d)
int main()
{
    int *d; (void) hipMalloc(&d, 256 * 8 * 256 * 4);
    RUN(add_u32); RUN(sub_u32); RUN(ashr); RUN(mad_i24); RUN(mul_i24); RUN(mad_u24); RUN(mul_lo); RUN(add_sdwa); RUN(bfe_i32);
    RUN(perm); RUN(med3); RUN(add3); RUN(lshl_add); RUN(add_lshl); RUN(lshl_or); RUN(fma_f32); RUN(mad_i32_i16);
    RUN(dot2_i32_i16); RUN(pk_add_i16); RUN(mul_hi_i32); RUN(mad_i64_i32); RUN(mov); RUN(cndmask); RUN(and_b32); RUN(lshlrev); RUN(max_i32); RUN(mul_i24_sdwa);
    RUN(mul_f32); RUN(floor_f32); RUN(cvt_f32_ubyte1); RUN(cvt_f32_ubyte2); RUN(cvt_i32_f32); RUN(cvt_f32_i32); RUN(cvt_pk_u8_f32);
    RUN(med3_f32); RUN(max_f32); RUN(dot4_i32_i8); RUN(fract_f32); RUN(sub_f32); RUN(xor_b32);
    return 0;
}
