// VALU issue-rate probe for gfx950: wave-instructions per cycle per SIMD for v_fma_f32, v_pk_fma_f32, v_fma_f64, v_mov_b32 dpp,
// v_cndmask, v_cvt_f64_f32 -- with 1, 2 and 4 wavefronts per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2 __attribute__((ext_vector_type(2)));
#define REP8(x) x x x x x x x x
template <int OP> __global__ void k(float *out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    v2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;
    const float m = 1.0000001f, c = 1e-9f;
    const v2 pm = {m, m}, pc = {c, c};
    const double dm = 1.0000001, dc = 1e-9;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));) }
        if (OP == 1) { REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pm), "v"(pc));) }
        if (OP == 2) { REP8(asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(dm), "v"(dc));) }
        if (OP == 3) { REP8(asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %5, %6 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %7, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (OP == 4) { REP8(asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(dc));) }
        if (OP == 5) { REP8(asm volatile("v_add_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));) }
        if (OP == 6) { REP8(asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %9\n v_pk_add_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %9\n v_pk_add_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %9\n v_pk_add_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %9" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pc), "v"(pm));) }
        if (OP == 7) { REP8(asm volatile("v_cvt_f64_f32 %0, %8\n v_cvt_f64_f32 %1, %9\n v_cvt_f64_f32 %2, %8\n v_cvt_f64_f32 %3, %9\n v_cvt_f64_f32 %4, %8\n v_cvt_f64_f32 %5, %9\n v_cvt_f64_f32 %6, %8\n v_cvt_f64_f32 %7, %9" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(a0), "v"(a1));) }
        if (OP == 8) { REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %9 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n v_pk_fma_f32 %1, %1, %8, %9 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n v_pk_fma_f32 %2, %2, %8, %9 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n v_pk_fma_f32 %3, %3, %8, %9 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n v_pk_fma_f32 %4, %4, %8, %9 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n v_pk_fma_f32 %5, %5, %8, %9 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n v_pk_fma_f32 %6, %6, %8, %9 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n v_pk_fma_f32 %7, %7, %8, %9 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pm), "v"(pc));) }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long *>(out)[(1 << 19) - 1] = t1 - t0;
}
template <int OP> void run(const char *name, float *d) {
    for (int wps : {1, 2, 4}) {
        const int iters = 20000;
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256 * wps), 0, 0, d, iters, 1.0f);  // warm-up
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256 * wps), 0, 0, d, iters, 1.0f);  // one workgroup per CU: wps wavefronts per SIMD
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        // wave-instructions per SIMD = wps * iters * 64; ns per wave-instruction per SIMD
        printf("%-28s waves/SIMD %d: kernel %.3f ms -> %.3f ns per wave-instruction per SIMD | ", name, wps, ms, ms * 1e6 / (wps * (double)iters * 64.0));
        long long ticks; hipMemcpy(&ticks, reinterpret_cast<long long *>(d) + (1 << 19) - 1, 8, hipMemcpyDeviceToHost);
        const double per = (double)ticks / (iters * 64.0);   // s_memtime ticks (100 MHz?) are converted below by the clock ratio estimate
        printf("%-28s waves/SIMD %d: %8.3f memtime ticks per wave-instruction (one wave's view)\n", name, wps, per);
    }
}
int main() {
    float *d; hipMalloc(&d, 1 << 22);
    run<0>("v_fma_f32", d); run<1>("v_pk_fma_f32", d); run<8>("v_pk_fma_f32 op_sel+neg", d); run<5>("v_add/mul_f32", d); run<6>("v_pk_add/mul_f32", d);
    run<2>("v_fma_f64", d); run<4>("v_add_f64", d); run<7>("v_cvt_f64_f32", d); run<3>("v_mov_b32_dpp row_shr", d);
    return 0;
}
