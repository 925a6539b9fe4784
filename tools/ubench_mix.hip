// Mixed-sequence VALU microbenchmark for gfx950 (MI355X): what a real in-order instruction stream gets out of the
// "2-cycle" simple ops (add/sub/ashr/mov) that tools/ubench_valu.hip measured in isolation.  Each kernel repeats one
// short instruction pattern (inline asm volatile: order and opcodes are exactly as written) and is run at 1, 2, 4 and
// 8 waves per SIMD.  Output: SIMD cycles per wave64 instruction, normalised to 2.4 GHz.
//     hipcc --offload-arch=gfx950 -O3 -o tools/ubench_mix.bin tools/ubench_mix.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITERS 1024

#define SUB(d, a, b)  asm volatile("v_sub_u32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b))
#define ADD(d, a, b)  asm volatile("v_add_u32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b))
#define MUL(d, a, b)  asm volatile("v_mul_i32_i24 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b))
#define ASHR(d, a)    asm volatile("v_ashrrev_i32 %0, 11, %1" : "=v"(d) : "v"(a))
#define MOVK(d)       asm volatile("v_mov_b32 %0, 0x80000000" : "=v"(d))
#define MED3(d, a, lo, hi) asm volatile("v_med3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(lo), "v"(hi))

// three one-pole low-passes, channel after channel (the order the compiler emits for k_active)
__global__ void k_iir3_seq(int *out, int c, int u)
{
    int h0 = threadIdx.x, h1 = h0 + 1, h2 = h0 + 2, t, vc = c, vu = u + (threadIdx.x & 3);
    for (int i = 0; i < ITERS; i++) {
        SUB(t, vu, h0); MUL(t, t, vc); ASHR(t, t); ADD(h0, h0, t);
        SUB(t, vu, h1); MUL(t, t, vc); ASHR(t, t); ADD(h1, h1, t);
        SUB(t, vu, h2); MUL(t, t, vc); ASHR(t, t); ADD(h2, h2, t);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = h0 ^ h1 ^ h2;
}
// the same 12 instructions, grouped by opcode
__global__ void k_iir3_grp(int *out, int c, int u)
{
    int h0 = threadIdx.x, h1 = h0 + 1, h2 = h0 + 2, t0, t1, t2, vc = c, vu = u + (threadIdx.x & 3);
    for (int i = 0; i < ITERS; i++) {
        SUB(t0, vu, h0); SUB(t1, vu, h1); SUB(t2, vu, h2);
        MUL(t0, t0, vc); MUL(t1, t1, vc); MUL(t2, t2, vc);
        ASHR(t0, t0); ASHR(t1, t1); ASHR(t2, t2);
        ADD(h0, h0, t0); ADD(h1, h1, t1); ADD(h2, h2, t2);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = h0 ^ h1 ^ h2;
}
// one dependent chain of a simple op / of a multiply: issue-to-issue latency of a lone dependency chain
__global__ void k_add_dep(int *out, int c, int u)
{
    int h = threadIdx.x, vc = c;
    for (int i = 0; i < ITERS; i++) {
        ADD(h, h, vc); ADD(h, h, vc); ADD(h, h, vc); ADD(h, h, vc); ADD(h, h, vc); ADD(h, h, vc);
        ADD(h, h, vc); ADD(h, h, vc); ADD(h, h, vc); ADD(h, h, vc); ADD(h, h, vc); ADD(h, h, vc);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = h;
}
__global__ void k_mul_dep(int *out, int c, int u)
{
    int h = threadIdx.x, vc = c;
    for (int i = 0; i < ITERS; i++) {
        MUL(h, h, vc); MUL(h, h, vc); MUL(h, h, vc); MUL(h, h, vc); MUL(h, h, vc); MUL(h, h, vc);
        MUL(h, h, vc); MUL(h, h, vc); MUL(h, h, vc); MUL(h, h, vc); MUL(h, h, vc); MUL(h, h, vc);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = h;
}
// 12 independent adds / 12 independent multiplies per iteration
__global__ void k_add_ind(int *out, int c, int u)
{
    int h0 = threadIdx.x, h1 = h0 + 1, h2 = h0 + 2, h3 = h0 + 3, h4 = h0 + 4, h5 = h0 + 5, vc = c;
    for (int i = 0; i < ITERS; i++) {
        ADD(h0, h0, vc); ADD(h1, h1, vc); ADD(h2, h2, vc); ADD(h3, h3, vc); ADD(h4, h4, vc); ADD(h5, h5, vc);
        ADD(h0, h0, vc); ADD(h1, h1, vc); ADD(h2, h2, vc); ADD(h3, h3, vc); ADD(h4, h4, vc); ADD(h5, h5, vc);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = h0 ^ h1 ^ h2 ^ h3 ^ h4 ^ h5;
}
__global__ void k_mul_ind(int *out, int c, int u)
{
    int h0 = threadIdx.x, h1 = h0 + 1, h2 = h0 + 2, h3 = h0 + 3, h4 = h0 + 4, h5 = h0 + 5, vc = c;
    for (int i = 0; i < ITERS; i++) {
        MUL(h0, h0, vc); MUL(h1, h1, vc); MUL(h2, h2, vc); MUL(h3, h3, vc); MUL(h4, h4, vc); MUL(h5, h5, vc);
        MUL(h0, h0, vc); MUL(h1, h1, vc); MUL(h2, h2, vc); MUL(h3, h3, vc); MUL(h4, h4, vc); MUL(h5, h5, vc);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = h0 ^ h1 ^ h2 ^ h3 ^ h4 ^ h5;
}
// independent simple op and multiply alternating: add, mul, add, mul ...
__global__ void k_alt_ind(int *out, int c, int u)
{
    int h0 = threadIdx.x, h1 = h0 + 1, h2 = h0 + 2, h3 = h0 + 3, h4 = h0 + 4, h5 = h0 + 5, vc = c;
    for (int i = 0; i < ITERS; i++) {
        ADD(h0, h0, vc); MUL(h1, h1, vc); ADD(h2, h2, vc); MUL(h3, h3, vc); ADD(h4, h4, vc); MUL(h5, h5, vc);
        ADD(h0, h0, vc); MUL(h1, h1, vc); ADD(h2, h2, vc); MUL(h3, h3, vc); ADD(h4, h4, vc); MUL(h5, h5, vc);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = h0 ^ h1 ^ h2 ^ h3 ^ h4 ^ h5;
}
// the same 6 adds + 6 multiplies as pairs: add, add, mul, mul ...
__global__ void k_pair_ind(int *out, int c, int u)
{
    int h0 = threadIdx.x, h1 = h0 + 1, h2 = h0 + 2, h3 = h0 + 3, h4 = h0 + 4, h5 = h0 + 5, vc = c;
    for (int i = 0; i < ITERS; i++) {
        ADD(h0, h0, vc); ADD(h2, h2, vc); MUL(h1, h1, vc); MUL(h3, h3, vc); ADD(h4, h4, vc); ADD(h0, h0, vc);
        MUL(h5, h5, vc); MUL(h1, h1, vc); ADD(h2, h2, vc); ADD(h4, h4, vc); MUL(h3, h3, vc); MUL(h5, h5, vc);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = h0 ^ h1 ^ h2 ^ h3 ^ h4 ^ h5;
}
// decoder filter stage (crt_decode.hip, tier 0): sub, mov (re-arm), v_mad_i64_i32 -- 4 cascaded stages, as emitted
#define MAD64(x, d, m) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(x) : "v"(d), "v"(m) : "vcc")
__global__ void k_stage64_seq(int *out, int c, int u)
{
    long x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    int d, lo, vc = c, vu = u + (threadIdx.x & 3);
    for (int i = 0; i < ITERS; i++) {
        SUB(d, vu, (int) (x0 >> 32)); MOVK(lo); x0 = (x0 & 0xffffffff00000000l) | (unsigned) lo; MAD64(x0, d, vc);
        SUB(d, (int) (x0 >> 32), (int) (x1 >> 32)); MOVK(lo); x1 = (x1 & 0xffffffff00000000l) | (unsigned) lo; MAD64(x1, d, vc);
        SUB(d, (int) (x1 >> 32), (int) (x2 >> 32)); MOVK(lo); x2 = (x2 & 0xffffffff00000000l) | (unsigned) lo; MAD64(x2, d, vc);
        SUB(d, (int) (x2 >> 32), (int) (x3 >> 32)); MOVK(lo); x3 = (x3 & 0xffffffff00000000l) | (unsigned) lo; MAD64(x3, d, vc);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (int) ((x0 ^ x1 ^ x2 ^ x3) >> 32);
}

// four INDEPENDENT stages (the luma-low, luma-high, I and Q cascades of one sample), grouped by opcode
__global__ void k_stage64_grp(int *out, int c, int u)
{
    long x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    int d0, d1, d2, d3, l0, l1, l2, l3, vc = c, vu = u + (threadIdx.x & 3);
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int k = 0; k < 1; k++) {
            SUB(d0, vu, (int) (x0 >> 32)); SUB(d1, vu, (int) (x1 >> 32)); SUB(d2, vu, (int) (x2 >> 32)); SUB(d3, vu, (int) (x3 >> 32));
            MOVK(l0); MOVK(l1); MOVK(l2); MOVK(l3);
            x0 = (x0 & 0xffffffff00000000l) | (unsigned) l0; x1 = (x1 & 0xffffffff00000000l) | (unsigned) l1;
            x2 = (x2 & 0xffffffff00000000l) | (unsigned) l2; x3 = (x3 & 0xffffffff00000000l) | (unsigned) l3;
            MAD64(x0, d0, vc); MAD64(x1, d1, vc); MAD64(x2, d2, vc); MAD64(x3, d3, vc);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (int) ((x0 ^ x1 ^ x2 ^ x3) >> 32);
}
// four independent stages, stage after stage (sub, mov, mad per stage)
__global__ void k_stage64_ind(int *out, int c, int u)
{
    long x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    int d, lo, vc = c, vu = u + (threadIdx.x & 3);
    for (int i = 0; i < ITERS; i++) {
        SUB(d, vu, (int) (x0 >> 32)); MOVK(lo); x0 = (x0 & 0xffffffff00000000l) | (unsigned) lo; MAD64(x0, d, vc);
        SUB(d, vu, (int) (x1 >> 32)); MOVK(lo); x1 = (x1 & 0xffffffff00000000l) | (unsigned) lo; MAD64(x1, d, vc);
        SUB(d, vu, (int) (x2 >> 32)); MOVK(lo); x2 = (x2 & 0xffffffff00000000l) | (unsigned) lo; MAD64(x2, d, vc);
        SUB(d, vu, (int) (x3 >> 32)); MOVK(lo); x3 = (x3 & 0xffffffff00000000l) | (unsigned) lo; MAD64(x3, d, vc);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (int) ((x0 ^ x1 ^ x2 ^ x3) >> 32);
}

// round 4: the same four filter stages on PACKED 16-bit state -- two independent cascades (a, b) share the registers X0 = {a0, b0},
// X1 = {a1, b1}; a stage pair is v_pk_sub_i16, two v_mad_i32_i16 (op_sel picks the half; the addend carries the rounding constant),
// v_perm_b32 of the two high words, v_pk_add_i16: 10 instructions for what stage64_* does in 12
#define PKSUB(d, a, b) asm volatile("v_pk_sub_i16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b))
#define PKADD(d, a, b) asm volatile("v_pk_add_i16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b))
#define MAD16LO(d, a, c, k) asm volatile("v_mad_i32_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(c), "v"(k))
#define MAD16HI(d, a, c, k) asm volatile("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(d) : "v"(a), "v"(c), "v"(k))
#define PERMHI(d, b, a, sel) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(d) : "v"(b), "v"(a), "v"(sel))
__global__ void k_stage16_pk(int *out, int c, int u)
{
    int x0 = threadIdx.x, x1 = x0 + 1, d, ra, rb, t, vc = c, vk = 32768, sel = 0x07060302, vu = u + (threadIdx.x & 3);
    for (int i = 0; i < ITERS; i++) {
        // two iterations' worth per loop trip would change nothing: the chain X0 -> X1 is the dependency that counts
        PKSUB(d, vu, x0); MAD16LO(ra, d, vc, vk); MAD16HI(rb, d, vc, vk); PERMHI(t, rb, ra, sel); PKADD(x0, x0, t);
        PKSUB(d, x0, x1); MAD16LO(ra, d, vc, vk); MAD16HI(rb, d, vc, vk); PERMHI(t, rb, ra, sel); PKADD(x1, x1, t);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1;
}
// ... and the form with SDWA adds writing the halves directly (no permute): pk_sub, 2 x mad, 2 x v_add_u32_sdwa
#define ADDHI_LO(x, r) asm volatile("v_add_u32_sdwa %0, %0, sext(%1) dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_1" : "+v"(x) : "v"(r))
#define ADDHI_HI(x, r) asm volatile("v_add_u32_sdwa %0, %0, sext(%1) dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(x) : "v"(r))
__global__ void k_stage16_sdwa(int *out, int c, int u)
{
    int x0 = threadIdx.x, x1 = x0 + 1, d, ra, rb, vc = c, vk = 32768, vu = u + (threadIdx.x & 3);
    for (int i = 0; i < ITERS; i++) {
        PKSUB(d, vu, x0); MAD16LO(ra, d, vc, vk); MAD16HI(rb, d, vc, vk); ADDHI_LO(x0, ra); ADDHI_HI(x0, rb);
        PKSUB(d, x0, x1); MAD16LO(ra, d, vc, vk); MAD16HI(rb, d, vc, vk); ADDHI_LO(x1, ra); ADDHI_HI(x1, rb);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1;
}
// the dot2 form of ONE cascade on {input, state} pairs: v_dot2_i32_i16 (c * u - c * x + K in one instruction), two SDWA adds
#define DOT2(d, a, c, k) asm volatile("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(c), "v"(k))
__global__ void k_stage16_dot2(int *out, int c, int u)
{
    int r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = 0, d, vc = (c & 0xffff) | (-c << 16), vk = 32768;
    for (int i = 0; i < ITERS; i++) {
        DOT2(d, r0, vc, vk); ADDHI_HI(r0, d); ADDHI_LO(r1, d);
        DOT2(d, r1, vc, vk); ADDHI_HI(r1, d); ADDHI_LO(r2, d);
        DOT2(d, r2, vc, vk); ADDHI_HI(r2, d); ADDHI_LO(r3, d);
        DOT2(d, r3, vc, vk); ADDHI_HI(r3, d); ADDHI_LO(r4, d);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4;
}

template <class K> void run(const char *name, K kern, int *d, int per_iter)
{
    hipEvent_t e0, e1; (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
    printf("%-14s", name);
    for (int wps = 1; wps <= 8; wps *= 2) {
        const int blocks = 256 * wps, threads = 256;                 // one wave per SIMD per block
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, 3, 5);
        (void) hipEventRecord(e0);
        for (int r = 0; r < 4; r++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, 3, 5);
        (void) hipEventRecord(e1); (void) hipEventSynchronize(e1);
        float ms; (void) hipEventElapsedTime(&ms, e0, e1); ms /= 4;
        const double winst = (double) blocks * (threads / 64) * ITERS * per_iter;
        printf("  %d w/SIMD: %5.2f", wps, 2.4e9 * (ms * 1e-3) * 1024.0 / winst);
    }
    printf("   (cycles per wave64 instr @2.4 GHz, %d instr/iter)\n", per_iter);
}
int main()
{
    int *d; (void) hipMalloc(&d, 256 * 8 * 256 * sizeof(int));
    run("iir3_seq", k_iir3_seq, d, 12);
    run("iir3_grp", k_iir3_grp, d, 12);
    run("add_dep", k_add_dep, d, 12);
    run("mul_dep", k_mul_dep, d, 12);
    run("add_ind", k_add_ind, d, 12);
    run("mul_ind", k_mul_ind, d, 12);
    run("alt_ind", k_alt_ind, d, 12);
    run("pair_ind", k_pair_ind, d, 12);
    run("stage64_seq", k_stage64_seq, d, 12);
    run("stage64_ind", k_stage64_ind, d, 12);
    run("stage64_grp", k_stage64_grp, d, 12);
    run("stage16_pk", k_stage16_pk, d, 10);          // 4 stages in 10 instructions: multiply cycles/instr by 10 (vs 12 for stage64_*)
    run("stage16_sdwa", k_stage16_sdwa, d, 10);
    run("stage16_dot2", k_stage16_dot2, d, 12);
    return 0;
}
