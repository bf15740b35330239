// Microbenchmark: sustained issue rate of wave64 VALU instructions on gfx950 at 1, 2, 4 waves per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 scripts/micro/valu_rate.hip -o /tmp/valu_rate ; run: /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int kIter = 4096;
// 8 independent accumulators per lane, 8 instructions per loop body (x4 unrolled by hand through the asm block)
#define BODY_F32 "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n" \
                 "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
#define BODY_MUL32 "v_mul_f32 %0, %0, %8\n v_add_f32 %1, %1, %9\n v_mul_f32 %2, %2, %8\n v_add_f32 %3, %3, %9\n" \
                   "v_mul_f32 %4, %4, %8\n v_add_f32 %5, %5, %9\n v_mul_f32 %6, %6, %8\n v_add_f32 %7, %7, %9\n"
#define BODY_F64 "v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n" \
                 "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
#define BODY_MUL64 "v_mul_f64 %0, %0, %8\n v_add_f64 %1, %1, %9\n v_mul_f64 %2, %2, %8\n v_add_f64 %3, %3, %9\n" \
                   "v_mul_f64 %4, %4, %8\n v_add_f64 %5, %5, %9\n v_mul_f64 %6, %6, %8\n v_add_f64 %7, %7, %9\n"
#define BODY_PK32 "v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n" \
                  "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
#define BODY_I32 "v_add_u32 %0, %0, %8\n v_and_b32 %1, %1, %9\n v_add_u32 %2, %2, %8\n v_xor_b32 %3, %3, %9\n" \
                 "v_add_u32 %4, %4, %8\n v_or_b32 %5, %5, %9\n v_add_u32 %6, %6, %8\n v_lshlrev_b32 %7, 1, %7\n"

template <typename T, int K>
__global__ void __launch_bounds__(256) k_rate(T *out, T a, T b, unsigned long long *cyc) {
    T x0 = (T)threadIdx.x, x1 = x0 + (T)1, x2 = x0 + (T)2, x3 = x0 + (T)3, x4 = x0 + (T)4, x5 = x0 + (T)5, x6 = x0 + (T)6, x7 = x0 + (T)7;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < kIter; ++i) {
        if constexpr (K == 0) asm volatile(BODY_F32 BODY_F32 BODY_F32 BODY_F32 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        if constexpr (K == 1) asm volatile(BODY_MUL32 BODY_MUL32 BODY_MUL32 BODY_MUL32 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        if constexpr (K == 2) asm volatile(BODY_F64 BODY_F64 BODY_F64 BODY_F64 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        if constexpr (K == 3) asm volatile(BODY_MUL64 BODY_MUL64 BODY_MUL64 BODY_MUL64 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        if constexpr (K == 4) asm volatile(BODY_PK32 BODY_PK32 BODY_PK32 BODY_PK32 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        if constexpr (K == 5) asm volatile(BODY_I32 BODY_I32 BODY_I32 BODY_I32 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = ((x0 + x1) + (x2 + x3)) + ((x4 + x5) + (x6 + x7));
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <typename T, int K>
void run(const char *name, int n_cu) {
    T *out; unsigned long long *cyc;
    hipMalloc(&out, sizeof(T) * 256 * n_cu * 8);
    hipMalloc(&cyc, 8);
    for (int wps : {1, 2, 4, 8}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k_rate<T, K><<<n_cu * wps, 256>>>(out, (T)1, (T)0, cyc);   // warm-up
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k_rate<T, K><<<n_cu * wps, 256>>>(out, (T)1, (T)0, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double insts_per_simd = (double)kIter * 32 * wps;   // wave-instructions issued on each SIMD
        printf("%-22s waves/SIMD %d: %7.3f ms  -> %.2f ns per wave-instruction per SIMD = %.2f cycles at 2.4 GHz; s_memtime cycles of block 0 per instruction of ITS wave: %.2f\n",
               name, wps, ms, ms * 1e6 / insts_per_simd, ms * 1e6 / insts_per_simd * 2.4, (double)c / (kIter * 32.0));
    }
    hipFree(out); hipFree(cyc);
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int n_cu = p.multiProcessorCount;
    printf("%s, %d CUs, clock %d MHz\n", p.gcnArchName, n_cu, p.clockRate / 1000);
    run<float, 0>("v_fma_f32", n_cu);
    run<float, 1>("v_mul_f32/v_add_f32", n_cu);
    run<double, 2>("v_fma_f64", n_cu);
    run<double, 3>("v_mul_f64/v_add_f64", n_cu);
    run<double, 4>("v_pk_fma_f32", n_cu);
    run<int, 5>("int32 add/logic", n_cu);
    return 0;
}
