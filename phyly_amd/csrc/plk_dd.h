/*
 * plk_dd.h -- double-double helpers (device + host), used by the exp(Qt)
 * kernel and by the deterministic weighted reductions.  A dd value is the
 * unevaluated sum hi + lo of two fp64 numbers with |lo| <= ulp(hi)/2.
 */
#ifndef PLK_DD_H
#define PLK_DD_H

#include <hip/hip_runtime.h>

struct dd { double hi, lo; };

__host__ __device__ static inline dd dd_make(double hi, double lo) { dd r; r.hi = hi; r.lo = lo; return r; }

__host__ __device__ static inline dd dd_two_sum(double a, double b)
{
    double s = a + b;
    double bb = s - a;
    double e = (a - (s - bb)) + (b - bb);
    return dd_make(s, e);
}

__host__ __device__ static inline dd dd_quick_two_sum(double a, double b)
{
    double s = a + b;
    double e = b - (s - a);
    return dd_make(s, e);
}

__host__ __device__ static inline dd dd_two_prod(double a, double b)
{
    double p = a * b;
    double e = fma(a, b, -p);
    return dd_make(p, e);
}

__host__ __device__ static inline dd dd_add(dd x, dd y)
{
    dd s = dd_two_sum(x.hi, y.hi);
    dd t = dd_two_sum(x.lo, y.lo);
    s.lo += t.hi;
    s = dd_quick_two_sum(s.hi, s.lo);
    s.lo += t.lo;
    return dd_quick_two_sum(s.hi, s.lo);
}

__host__ __device__ static inline dd dd_add_d(dd x, double y)
{
    dd s = dd_two_sum(x.hi, y);
    s.lo += x.lo;
    return dd_quick_two_sum(s.hi, s.lo);
}

__host__ __device__ static inline dd dd_mul(dd x, dd y)
{
    dd p = dd_two_prod(x.hi, y.hi);
    p.lo += x.hi * y.lo + x.lo * y.hi;
    return dd_quick_two_sum(p.hi, p.lo);
}

__host__ __device__ static inline dd dd_mul_d(dd x, double y)
{
    dd p = dd_two_prod(x.hi, y);
    p.lo += x.lo * y;
    return dd_quick_two_sum(p.hi, p.lo);
}

/* x * 2^e, exact */
__host__ __device__ static inline dd dd_ldexp(dd x, int e)
{
    return dd_make(ldexp(x.hi, e), ldexp(x.lo, e));
}

#endif
