// Which of HIP's fp32 divide / square-root spellings are correctly rounded on gfx950?  (rmsnorm_quantize.hip: rvar = 1 / sqrt(s / K + eps).)
// Measured: __fdiv_rn is, __builtin_sqrtf is, __fsqrt_rn is NOT (15 % of inputs one ulp off: it is the native v_sqrt_f32).
// hipcc --offload-arch=gfx950 -O3 tools/probe_rvar.hip -o /tmp/probe_rvar && /tmp/probe_rvar
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
// (a kernel of its own: next to __fsqrt_rn of the same argument the two calls are merged into the less precise one)
__global__ void k_sqrt(const float *s, int n, float K, float eps, float *d) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d[i] = __builtin_sqrtf(__fdiv_rn(s[i], K) + eps);   // llvm.sqrt.f32: correctly rounded under HIP's default -fhip-fp32-correctly-rounded-divide-sqrt
}
__global__ void k_rvar(const float *s, int n, float K, float eps, float *a, float *b, float *c) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float mean = __fdiv_rn(s[i], K);
    a[i] = mean;
    const float q = __fsqrt_rn(mean + eps);
    b[i] = q;
    c[i] = __fdiv_rn(1.0f, q);
}
int main() {
    const int n = 1 << 22;
    std::vector<float> s(n), a(n), b(n), c(n), d(n);
    srand(1);
    for (int i = 0; i < n; ++i) s[i] = (float)((double)rand() / RAND_MAX * 8192.0 + 1e-3) * (i % 3 == 0 ? 1e-4f : 1.0f);
    float *ds, *da, *db, *dc, *dd;
    hipMalloc(&ds, n * 4); hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dc, n * 4); hipMalloc(&dd, n * 4);
    hipMemcpy(ds, s.data(), n * 4, hipMemcpyHostToDevice);
    const float K = 4096.0f, eps = 1e-5f;
    hipLaunchKernelGGL(k_rvar, dim3(n / 256), dim3(256), 0, 0, ds, n, K, eps, da, db, dc);
    hipLaunchKernelGGL(k_sqrt, dim3(n / 256), dim3(256), 0, 0, ds, n, K, eps, dd);
    hipMemcpy(a.data(), da, n * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), db, n * 4, hipMemcpyDeviceToHost); hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost); hipMemcpy(d.data(), dd, n * 4, hipMemcpyDeviceToHost);
    long bad_div = 0, bad_sqrt = 0, bad_rcp = 0, bad_bsqrt = 0, bad_host = 0;
    for (int i = 0; i < n; ++i) {
        const float mean = s[i] / K;                       // IEEE on the host
        if (mean != a[i]) ++bad_div;
        const float arg = a[i] + eps;
        const float q = (float)sqrt((double)arg);          // correctly rounded (double sqrt of a float, rounded once more, is exact enough)
        if (sqrtf(arg) != q) ++bad_host;
        if (q != b[i]) ++bad_sqrt;
        if (q != d[i]) ++bad_bsqrt;
        const float r = 1.0f / b[i];
        if (r != c[i]) ++bad_rcp;
    }
    printf("%d values: __fdiv_rn(s, K) wrong %ld, __fsqrt_rn wrong %ld, __builtin_sqrtf wrong %ld, __fdiv_rn(1, q) wrong %ld (host sqrtf wrong %ld)\n", n, bad_div, bad_sqrt, bad_bsqrt, bad_rcp, bad_host);
    return 0;
}
