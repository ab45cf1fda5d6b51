// Hardware probe for the packed-FP32 instructions of gfx950: does an instruction whose destination pair overlaps a source
// pair still see the OLD source in its second (high) half?  (hipcc --offload-arch=gfx950 -O3 tools/lab/pk_probe.cpp -o tools/pk_probe)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float cf __attribute__((ext_vector_type(2)));
__global__ void k(const cf *a, const cf *b, cf *out) {
    const int i = threadIdx.x;
    cf x = a[i], y = b[i];
    // 1: plain in-place add (dst = src1)
    { cf d = y; asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(d) : "v"(x)); out[8 * i + 0] = d; }
    // 2: swizzled add, dst = src1: (x.x + y.y, x.y - y.x)
    { cf d = y; asm volatile("v_pk_add_f32 %0, %1, %0 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,0] neg_hi:[0,1]" : "+v"(d) : "v"(x)); out[8 * i + 1] = d; }
    // 3: swizzled add, dst = src0: (x.x + y.y, x.y - y.x) with x in place
    { cf d = x; asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,0] neg_hi:[0,1]" : "+v"(d) : "v"(y)); out[8 * i + 2] = d; }
    // 4: broadcast multiply, dst = the broadcast source: (x.x * y.x, x.y * y.x)
    { cf d = y; asm volatile("v_pk_mul_f32 %0, %1, %0 op_sel:[0,0] op_sel_hi:[1,0]" : "+v"(d) : "v"(x)); out[8 * i + 3] = d; }
    // 5: broadcast of the HIGH half, dst = that source: (x.x * y.y, x.y * y.y)
    { cf d = y; asm volatile("v_pk_mul_f32 %0, %1, %0 op_sel:[0,1] op_sel_hi:[1,1]" : "+v"(d) : "v"(x)); out[8 * i + 4] = d; }
    // 6: fma with the accumulator in place (no swizzle on it): x.yy * (-y.y, y.x) + d
    { cf d = x * y; asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0] neg_hi:[0,0,0]" : "+v"(d) : "v"(x), "v"(y)); out[8 * i + 5] = d; }
    // 7: fma, dst = src0 (used as x.yy): (x.y * -y.y + 1, x.y * y.x + 2)
    { cf d = x; cf t = cf{1.f, 2.f}; asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0] neg_hi:[0,0,0]" : "+v"(d) : "v"(y), "v"(t)); out[8 * i + 6] = d; }
    // 8: fma, dst = src1 (swizzled): same value
    { cf d = y; cf t = cf{1.f, 2.f}; asm volatile("v_pk_fma_f32 %0, %1, %0, %2 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0] neg_hi:[0,0,0]" : "+v"(d) : "v"(x), "v"(t)); out[8 * i + 7] = d; }
}
int main() {
    const int n = 64;
    cf ha[n], hb[n], ho[8 * n];
    for (int i = 0; i < n; ++i) { ha[i] = cf{1.5f + i, -2.25f + 0.5f * i}; hb[i] = cf{0.75f - i, 3.0f + 0.25f * i}; }
    cf *da, *db, *dout;
    hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dout, sizeof(ho));
    hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(n), 0, 0, da, db, dout);
    hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
    int bad[8] = {0};
    for (int i = 0; i < n; ++i) {
        const cf x = ha[i], y = hb[i];
        const cf want[8] = {x + y, cf{x.x + y.y, x.y - y.x}, cf{x.x + y.y, x.y - y.x}, cf{x.x * y.x, x.y * y.x}, cf{x.x * y.y, x.y * y.y},
                            cf{__builtin_fmaf(x.y, -y.y, x.x * y.x), __builtin_fmaf(x.y, y.x, x.y * y.y)},
                            cf{__builtin_fmaf(x.y, -y.y, 1.f), __builtin_fmaf(x.y, y.x, 2.f)}, cf{__builtin_fmaf(x.y, -y.y, 1.f), __builtin_fmaf(x.y, y.x, 2.f)}};
        for (int c = 0; c < 8; ++c) if (ho[8 * i + c].x != want[c].x || ho[8 * i + c].y != want[c].y) {
            if (!bad[c]) printf("case %d lane %d: got (%g, %g) want (%g, %g)\n", c + 1, i, ho[8 * i + c].x, ho[8 * i + c].y, want[c].x, want[c].y);
            bad[c]++;
        }
    }
    for (int c = 0; c < 8; ++c) printf("case %d: %s (%d of %d lanes differ)\n", c + 1, bad[c] ? "OVERLAP HAZARD" : "ok", bad[c], n);
    return 0;
}
