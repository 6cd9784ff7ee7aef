// Diagnostic (not part of the product or the test-suite): device sincos/exp accuracy vs host long double.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double *x, double *s, double *c, double *e, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { sincos(x[i], &s[i], &c[i]); e[i] = exp(x[i] * 1e-9); }
}
int main()
{
    const int n = 1 << 16;
    std::vector<double> x(n), s(n), c(n), e(n);
    for (int i = 0; i < n; i++) { double t = (double)i; x[i] = 0.5 * t * t * 6.98e-8 + 1e-3 * i; }
    double *dx, *ds, *dc, *de;
    hipMalloc(&dx, n * 8); hipMalloc(&ds, n * 8); hipMalloc(&dc, n * 8); hipMalloc(&de, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, ds, dc, de, n);
    hipMemcpy(s.data(), ds, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), dc, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(e.data(), de, n * 8, hipMemcpyDeviceToHost);
    double ms = 0, mc = 0, me = 0, mx = 0;
    for (int i = 0; i < n; i++) {
        long double xl = x[i];
        ms = fmax(ms, fabs((double)(sinl(xl) - (long double)s[i])));
        mc = fmax(mc, fabs((double)(cosl(xl) - (long double)c[i])));
        me = fmax(me, fabs((double)(expl(xl * 1e-9L) - (long double)e[i])));
        mx = fmax(mx, x[i]);
    }
    printf("max arg %.3f  max |sin err| %.3e  |cos err| %.3e  |exp err| %.3e\n", mx, ms, mc, me);
    return 0;
}
