"""Objectives written the way a CALLER writes them for the run-time compiled form (fl_user_compile, include/fl_nlopt.h): HIP
source text of a functor with the interface of include/fl_user_objective.hpp.  Test material: each restates one of the
library's built-in objectives, so the result can be held to the built-in kernel and to the oracle bit for bit."""

# f = 1/2 sum d x^2 - sum b x  (element-wise; data0 = d, data1 = b, params = one double `half`: every term of the first sum
# is shifted by half - 1/2 -- nothing at 0.5, a constant offset of f otherwise: shows that the parameter block arrives, with an
# objective that stays finite and consistent with its gradient.  NEVER test with a NaN objective: the reference's zoom loop
# does not terminate on NaN (NO.f90:1557-1579 compares its way out) and neither does its restatement -- a kernel that never ends)
DIAGQUAD = r"""
template <int NW, int EPT> struct MyQuadratic {
    static constexpr int LDS_DOUBLES = 0;
    double d[EPT], b[EPT], half;
    __device__ void init(const fl::SolveArgs &A, int prob, double *)
    {
        fl::load_user<NW, EPT>(A.d + (size_t)prob * A.n, A.n, d);
        fl::load_user<NW, EPT>(A.b + (size_t)prob * A.n, A.n, b);
        half = A.user ? *static_cast<const double *>(A.user) : 0.5;
    }
    __device__ void eval(const double (&x)[EPT], double (&g)[EPT], double &s0, double &s1, int, double *)
    {
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const double dx = d[k] * x[k];
            const double t0 = dx * x[k] + (half - 0.5), t1 = b[k] * x[k];
            g[k] = dx - b[k];
            s0 = (k == 0) ? t0 : s0 + t0;
            s1 = (k == 0) ? t1 : s1 + t1;
        }
    }
    __device__ static double combine(double s0, double s1) { return 0.5 * s0 - s1; }
};
"""

# chained Rosenbrock f = sum_{i<n-1} 100 (x_{i+1} - x_i^2)^2 + (1 - x_i)^2: NEIGHBOUR-COUPLED -- every thread needs the
# elements next to its 16-byte chunks, so x is staged through LDS (LDS_DOUBLES, two barriers per evaluation)
ROSENBROCK = r"""
template <int NW, int EPT> struct MyRosenbrock {
    using G = fl::Geo<NW, EPT>;
    static constexpr int LDS_DOUBLES = G::NPAD + 2;   // x with one halo element on each side
    __device__ void init(const fl::SolveArgs &, int, double *xs)
    {
        if (G::ltid() == 0) {
            xs[0] = 0.0;
            xs[G::NPAD + 1] = 0.0;
        }
    }
    __device__ void eval(const double (&x)[EPT], double (&g)[EPT], double &s0, double &s1, int n, double *xs)
    {
        s1 = 0.0;
        __syncthreads();                               // the previous evaluation's neighbour reads are complete
        for (int c = 0; c < G::NCH; ++c) {
            const int e = G::e0(c);                    // the chunk's first element
            xs[1 + e] = x[2 * c];
            xs[2 + e] = x[2 * c + 1];
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < G::NCH; ++c) {
            const int e = G::e0(c);
            const double xa = x[2 * c], xb = x[2 * c + 1];
            const double xl = xs[e], xr = xs[e + 3];   // x[e-1], x[e+2]
            const double ul = xa - xl * xl, ua = xb - xa * xa, ub = xr - xb * xb;
            const double va = 1.0 - xa, vb = 1.0 - xb;
            const double A_a = (e >= 1) ? 200.0 * ul : 0.0;
            const double A_b = 200.0 * ua;
            double ta = 0.0, tb = 0.0, ga = 0.0, gb = 0.0;
            if (e <= n - 2) {
                ta = 100.0 * (ua * ua) + va * va;
                ga = A_a - 400.0 * xa * ua - 2.0 * va;
            } else if (e == n - 1) {
                ga = A_a;
            }
            if (e + 1 <= n - 2) {
                tb = 100.0 * (ub * ub) + vb * vb;
                gb = A_b - 400.0 * xb * ub - 2.0 * vb;
            } else if (e + 1 == n - 1) {
                gb = A_b;
            }
            g[2 * c] = ga;
            g[2 * c + 1] = gb;
            s0 = (c == 0) ? ta : s0 + ta;
            s0 = s0 + tb;
        }
    }
    __device__ static double combine(double s0, double) { return s0; }
};
"""

BROKEN = r"""
template <int NW, int EPT> struct Oops {
    static constexpr int LDS_DOUBLES = 0;
    __device__ void init(const fl::SolveArgs &, int, double *) {}
    __device__ void eval(const double (&x)[EPT], double (&g)[EPT], double &s0, double &s1, int, double *) { s0 = undeclared_name; }
    __device__ static double combine(double s0, double) { return s0; }
};
"""

# ---- constraints of the caller's own (fl_user_compile_auglag with a constraints class; NO.f90:1928-1934: c, cd)
# M block spheres c_j = sum_{i in block j} x_i^2 - 1 over M consecutive blocks of n / M elements: the library's own family
# restated as a USER functor (held to the built-in kernel and to the oracle bit for bit)
BLOCK_SPHERES = r"""
template <int NW, int EPT> struct MySpheres {
    using G = fl::Geo<NW, EPT>;
    int w, m;
    __device__ void init(const fl::SolveArgs &A, int) { m = A.aug_m; w = A.n / A.aug_m; }
    __device__ void partial(const double (&x)[EPT], double (&cp)[8], int n)
    {
        for (int j = 0; j < 8; ++j) {
            double acc = 0.0;
            for (int k = 0; k < EPT; ++k) {
                const int e = G::e0(k >> 1) + (k & 1);
                const double t = (e < n && e / w == j) ? x[k] * x[k] : 0.0;
                acc = (k == 0) ? t : acc + t;
            }
            cp[j] = acc;
        }
    }
    __device__ double offset(int) const { return -1.0; }
    __device__ void add_gradient(const double (&x)[EPT], const double (&v)[8], double (&g)[EPT], int n)
    {
        for (int k = 0; k < EPT; ++k) {
            const int e = G::e0(k >> 1) + (k & 1);
            if (e < n) {
                double vv = 0.0;
                for (int j = 0; j < 8; ++j) vv = (e / w == j) ? v[j] : vv;
                g[k] = g[k] + (2.0 * x[k]) * vv;
            }
        }
    }
};
"""

# M LINEAR constraints over interleaved index sets: c_j = sum_{i = j mod M} x_i - 1 (any n; nothing like the built-in family)
STRIDE_SUMS = r"""
template <int NW, int EPT> struct MyStrideSums {
    using G = fl::Geo<NW, EPT>;
    int m;
    __device__ void init(const fl::SolveArgs &A, int) { m = A.aug_m; }
    __device__ void partial(const double (&x)[EPT], double (&cp)[8], int n)
    {
        for (int j = 0; j < 8; ++j) {
            double acc = 0.0;
            for (int k = 0; k < EPT; ++k) {
                const int e = G::e0(k >> 1) + (k & 1);
                acc = acc + ((e < n && e % m == j) ? x[k] : 0.0);
            }
            cp[j] = acc;
        }
    }
    __device__ double offset(int) const { return -1.0; }
    __device__ void add_gradient(const double (&x)[EPT], const double (&v)[8], double (&g)[EPT], int n)
    {
        for (int k = 0; k < EPT; ++k) {
            const int e = G::e0(k >> 1) + (k & 1);
            if (e < n) {
                double vv = 0.0;
                for (int j = 0; j < 8; ++j) vv = (e % m == j) ? v[j] : vv;
                g[k] = g[k] + vv;
            }
        }
    }
};
"""

# ---- n > 4096: the STREAMING functor (fortran-library_amd/csrc/fl_big.hpp): a plain class asked one element pair at a time
# the diagonal quadratic again (data0 = d, data1 = b), restating the built-in vectors-in-HBM kernel's arithmetic
STREAM_DIAGQUAD = r"""
struct MyBigQuadratic {
    static constexpr bool NEIGHBOURS = false;
    const double *d, *b;
    __device__ void init(const fl::SolveArgs &A, int prob)
    {
        d = A.d + (size_t)prob * A.n;
        b = A.b + (size_t)prob * A.n;
    }
    __device__ void pair(int e, int n, const double *, double xa, double xb, double &ta, double &tb, double &ua, double &ub,
                         double &ga, double &gb)
    {
        const double da = e < n ? d[e] : 0.0, db = e + 1 < n ? d[e + 1] : 0.0;
        const double ba = e < n ? b[e] : 0.0, bb = e + 1 < n ? b[e + 1] : 0.0;
        const double dxa = da * xa, dxb = db * xb;
        ta = dxa * xa;
        tb = dxb * xb;
        ua = ba * xa;
        ub = bb * xb;
        ga = dxa - ba;
        gb = dxb - bb;
    }
    __device__ static double combine(double s0, double s1) { return 0.5 * s0 - s1; }
};
"""

# chained Rosenbrock: NEIGHBOUR-COUPLED -- pair() reads x[e-1] and x[e+2] from the row (NEIGHBOURS = true: the trial point is
# stored in a pass of its own and a barrier precedes the evaluation)
STREAM_ROSENBROCK = r"""
struct MyBigRosenbrock {
    static constexpr bool NEIGHBOURS = true;
    __device__ void init(const fl::SolveArgs &, int) {}
    __device__ void pair(int e, int n, const double *x, double xa, double xb, double &ta, double &tb, double &ua, double &ub,
                         double &ga, double &gb)
    {
        const double xl = (e >= 1 && e - 1 < n) ? x[e - 1] : 0.0;
        const double xr = (e + 2 < n) ? x[e + 2] : 0.0;
        const double ul = xa - xl * xl, um = xb - xa * xa, ur = xr - xb * xb;
        const double va = 1.0 - xa, vb = 1.0 - xb;
        const double A_a = (e >= 1) ? 200.0 * ul : 0.0;
        const double A_b = 200.0 * um;
        ta = tb = ga = gb = ua = ub = 0.0;
        if (e <= n - 2) {
            ta = 100.0 * (um * um) + va * va;
            ga = A_a - 400.0 * xa * um - 2.0 * va;
        } else if (e == n - 1) {
            ga = A_a;
        }
        if (e + 1 <= n - 2) {
            tb = 100.0 * (ur * ur) + vb * vb;
            gb = A_b - 400.0 * xb * ur - 2.0 * vb;
        } else if (e + 1 == n - 1) {
            gb = A_b;
        }
    }
    __device__ static double combine(double s0, double) { return s0; }
};
"""
