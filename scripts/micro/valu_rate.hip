// valu_rate.hip -- diagnostic microbenchmark (not part of the product): sustained VALU issue rate per SIMD for plain
// Float32, packed Float32 and Float64 FMAs and for 32-bit integer/compare forms at 1, 2, 4 and 8 waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 -o gpurun_out/valu_rate scripts/micro/valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float float2v __attribute__((ext_vector_type(2)));
constexpr int kIter = 4096, kAcc = 8;

template <int KIND>
__global__ void k_rate(float *out, float seed) {
    float a[kAcc]; double d[kAcc]; float2v p[kAcc]; int q[kAcc];
    const float b = seed * 1.0001f, c = seed * 0.5f;
    const double bd = b, cd = c; const float2v bp = {b, b}, cp = {c, c};
    for (int i = 0; i < kAcc; ++i) { a[i] = threadIdx.x * seed + i; d[i] = a[i]; p[i] = (float2v){a[i], a[i]}; q[i] = (int)a[i]; }
    for (int it = 0; it < kIter; ++it) {
#pragma unroll
        for (int i = 0; i < kAcc; ++i) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(p[i]) : "v"(bp), "v"(cp));
            if (KIND == 2) asm volatile("v_fma_f64 %0, %1, %0, %2" : "+v"(d[i]) : "v"(bd), "v"(cd));
            if (KIND == 3) asm volatile("v_add_u32 %0, %1, %0" : "+v"(q[i]) : "v"(it));
            if (KIND == 4) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (KIND == 5) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(c));
            if (KIND == 6) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c) : "vcc");
            if (KIND == 7) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(c));
            if (KIND == 8) asm volatile("v_fma_f32 %0, |%1|, %0, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        }
    }
    float s = 0;
    for (int i = 0; i < kAcc; ++i) s += a[i] + (float)d[i] + p[i].x + p[i].y + q[i];
    if (s == 12345.678f) out[0] = s;
}

template <int KIND>
void run(const char *name, int instr_per_step, float *out) {
    for (int w : {1, 2, 4, 8}) {
        const int threads = 64 * 4 * (w > 4 ? 4 : w), blocks = 256 * (w > 4 ? w / 4 : 1) * 4;  // 4 rounds of full-chip blocks
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(threads), 0, 0, out, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(threads), 0, 0, out, 1.0f);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double waves = (double)blocks * threads / 64.0;
        const double instr = waves * kIter * kAcc * instr_per_step;
        const double per_simd_ns = instr / 1024.0 / (ms * 1e6);
        printf("%-28s waves/SIMD %d  %8.3f ms  %6.3f wave-instr/ns/SIMD  = %5.2f cycles/instr at 2.4 GHz\n", name, w, ms, per_simd_ns, 2.4 / per_simd_ns);
    }
}
int main() {
    float *out; hipMalloc(&out, 64);
    run<0>("v_fma_f32", 1, out);
    run<8>("v_fma_f32 |abs| modifier", 1, out);
    run<4>("v_mul_f32", 1, out);
    run<5>("v_add_f32", 1, out);
    run<1>("v_pk_fma_f32", 1, out);
    run<2>("v_fma_f64", 1, out);
    run<3>("v_add_u32", 1, out);
    run<7>("v_mov_b32", 1, out);
    run<6>("v_cmp_lt_f32+v_cndmask", 2, out);
    return 0;
}
