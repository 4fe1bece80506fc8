// fl_eig_vectors.hip -- My_dsyev('V', ...) (LinearAlgebra.f90:879-887: dsyev(jobtype,'L',...), eigenvalues ascending,
// A <- normalised eigenvectors) for one large matrix, laid out for the GPU:
//
//   1. Householder tridiagonalisation, one launch per reflector, reflectors kept in A (fl_blas_kernels.hip:
//      tridiag_step_kernel, the same launches as the 'N' job), and the tridiagonal's eigenvalues by multisection;
//   2. the tridiagonal's eigenvectors by INVERSE ITERATION, one lane per eigenvector (invit_kernel): T - sigma I is
//      factorised with partial pivoting (LAPACK's dlagtf elimination), three solves from a random start, close
//      eigenvalues get shifts 10 eps ||T|| apart (dstein's rule) -- n independent sequential recurrences of length n,
//      coalesced over the eigenvectors ([component][vector] layout), O(n^2) work in all;
//   3. orthogonality not by dstein's sequential Gram-Schmidt inside clusters but for ALL vectors at once on the matrix
//      cores: Cholesky-QR of the n x n block of vectors -- G = Y Y^T (dgemm), G = L L^T (blocked Cholesky), Y <- L^-1 Y.
//      Mixing two vectors whose eigenvalues differ by delta changes the residual by (their overlap ~ eps ||T|| / delta)
//      x delta: the residuals stay at the eps ||T|| level.  When G was far from the identity (clusters, multiple
//      eigenvalues: the vectors of a cluster come out as independent but ill-conditioned combinations) one more solve
//      from the whitened vectors and a second Cholesky-QR follow (see fl_dsyev_vectors);
//   4. back-transformation Z = H_0 H_1 ... H_{n-3} Y^T: a wave (n <= 2048; beyond, a workgroup) owns one eigenvector in
//      registers and applies every reflector to it -- the reflectors (n^2/2 doubles) are read from L2 by everyone, the
//      next one is in flight while the current one is applied;
//   5. the result is CHECKED on the device before it is returned: max |Y Y^T - I| and max |T y - lambda y| (O(n^2)); a
//      failed check (or a Gram matrix that is not positive definite) returns 1 and the caller falls back to cyclic
//      Jacobi (fl_dsyev_jacobi) -- the fast path cannot return a bad basis silently.
//
// n = 1024: 165 ms (Jacobi) -> see DESIGN.md 8 for the measured time.  Tests: tests/test_gpu_la_reference.py (the
// reference's own dsyev results for residual / orthogonality bars; identity, projector, clusters, Wilkinson, graded...).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/fl_nlopt.h"
#include "fl_host.hpp"
#include "fl_reduce.hpp"

extern "C" {
int fl_sytrd_values(int n, double *A_dev, double *w_dev, double *ws, double *tauvec, double *Vkeep, int ldk, void *stream);
int fl_chol_whiten(int n, int ncols, double *G_dev, int ldg, double *Y_dev, int32_t *info_dev, void *ws_dev, size_t ws_bytes,
                   void *stream);
size_t fl_chol_blocked_workspace_bytes(int batch, int n, int nrhs_tmp);
int fl_dgemm_strided(int transA, int transB, int M, int K, int N, double alpha, const double *A_dev, int lda, size_t strideA,
                     const double *B_dev, int ldb, size_t strideB, double beta, double *C_dev, int ldc, size_t strideC,
                     int batch, int lower_only, void *stream);
}

namespace fl {

constexpr double EV_EPS = 2.220446049250313e-16;

// max over non-negative doubles through their bit patterns (monotone); NaN counts as +inf
__device__ __forceinline__ void atomic_max_nonneg(double *addr, double v)
{
    if (!(v == v)) v = __builtin_inf();
    atomicMax(reinterpret_cast<unsigned long long *>(addr), (unsigned long long)__double_as_longlong(v));
}

// scal[0] = ||T||_inf, sigma = the shifts.  Eigenvalues closer than sep = 10 eps ||T|| to their neighbour form a cluster
// whose shifts are put sep apart (dstein's rule: equal shifts would make every vector of the cluster the same dominant
// combination).  dstein chains them upwards and orthogonalises as it goes; here nothing is orthogonalised until all
// vectors exist, so a LARGE cluster must not walk its shifts into the eigenvalues above it -- the null cluster of a
// numerically rank-deficient matrix (a Hilbert matrix: 370 of 400 eigenvalues within 1e-16 of zero, the next ones at
// 1e-14, 1e-13, ...) chained upwards put hundreds of shifts next to those, and hundreds of vectors collapsed onto theirs.
// A cluster is therefore chained to the side where it has the room: upwards if the eigenvalue above it stays more than
// sep beyond its last shift, else downwards (below the previous cluster's shifts), else upwards as before; and members
// that are numerically the same eigenvalue share one shift beyond their run instead of taking one sep each (below).
// One workgroup of 256; dyn. LDS 2 n doubles; the scan over the clusters is sequential (thread 0, in LDS).
__global__ __launch_bounds__(256) void invit_shift_kernel(int n, const double *w, const double *d, const double *e,
                                                          const double *last_diag, double *sigma, double *scal)
{
    extern __shared__ double sh[];
    double *ws = sh, *sg = sh + n;
    __shared__ double red[256];
    const int tid = threadIdx.x;
    double tn = 0.0;
    for (int i = tid; i < n; i += 256) {
        const double di = (i == n - 1) ? last_diag[0] : d[i];
        const double el = (i > 0) ? fabs(e[i - 1]) : 0.0, er = (i < n - 1) ? fabs(e[i]) : 0.0;
        tn = fmax(tn, fabs(di) + el + er);
        ws[i] = w[i];
    }
    red[tid] = tn;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) red[tid] = fmax(red[tid], red[tid + o]);
        __syncthreads();
    }
    tn = red[0];
    if (tid == 0) {
        scal[0] = tn;
        const double sep = 10.0 * EV_EPS * tn, tight = 0.05 * sep;
        // One cluster i .. j, chained in direction dir (+1 upwards, -1 downwards: the mirror image, values negated and
        // taken from the top).  Members that are TIGHT to their neighbour (closer than eps ||T|| / 2: numerically the same
        // eigenvalue) do not take one sep each: the first of such a run keeps the chain's shift, the others SHARE one
        // shift beyond the run's far end, by D = max(sep, the run's width) -- from there their weights in the solves differ
        // by at most 2^3, the random starts alone keep the vectors independent, and the run blurs its surroundings by D
        // only (chained one sep each, 370 null eigenvalues reach 2e-12 ||T|| away and take the genuine small ones in;
        // sharing a shift INSIDE the run's width makes every vector the combination nearest to it).
        // Returns the last shift (in the mirrored coordinate for dir = -1); write = false: a dry run.
        auto chain = [&](int i, int j, int dir, bool write) {
            const int c = j - i + 1;
            auto v = [&](int k) { return dir > 0 ? ws[i + k] : -ws[j - k]; };
            auto set = [&](int k, double x) {
                if (write) sg[dir > 0 ? i + k : j - k] = dir > 0 ? x : -x;
            };
            double prev = v(0);
            set(0, prev);
            int k = 1;
            while (k < c) {
                if (v(k) - v(k - 1) < tight) {
                    int r = k;
                    while (r + 1 < c && v(r + 1) - v(r) < tight) ++r; // the run k-1 .. r
                    const double shared = fmax(v(r), prev) + fmax(sep, v(r) - v(k - 1));
                    for (int q = k; q <= r; ++q) set(q, shared);
                    prev = shared;
                    k = r + 1;
                } else {
                    prev = fmax(v(k), prev + sep);
                    set(k, prev);
                    ++k;
                }
            }
            return prev;
        };
        int i = 0;
        while (i < n) {
            int j = i;
            while (j + 1 < n && ws[j + 1] - ws[j] < sep) ++j; // the cluster i .. j
            int dir = 1;
            if (j > i && j + 1 < n && chain(i, j, 1, false) + sep > ws[j + 1]) { // upwards it would reach the eigenvalue above
                const double lowest = -chain(i, j, -1, false);
                if (i == 0 || lowest - sep >= sg[i - 1]) dir = -1;
            }
            chain(i, j, dir, true);
            i = j + 1;
        }
    }
    __syncthreads();
    for (int i = tid; i < n; i += 256) sigma[i] = sg[i];
}

__device__ __forceinline__ double invit_start(unsigned v, unsigned k) // uniform in (-1, 1), a hash of (vector, component)
{
    unsigned long long z = (unsigned long long)v * 0x9E3779B97F4A7C15ull ^ ((unsigned long long)k * 0xBF58476D1CE4E5B9ull + 0x94D049BB133111EBull);
    z ^= z >> 30;
    z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27;
    z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return 2.0 * ((double)(z >> 11) * 0x1.0p-53) - 1.0;
}

// One lane per eigenvector: factorisation of T - sigma_v I with partial pivoting (rows k, k+1 compared like dlagtf
// without its scaling), three solves, normalisation.  All arrays [k][v] (ld n): F_ra = reciprocals of the (perturbed)
// pivots, F_b / F_d2 = first and second superdiagonal of U, F_c = multipliers, F_sw = row interchanges; Y [component][vector].
// T is scaled by a power of two to ||T|| in [1/2, 1) (exact), so that nothing here depends on the matrix's magnitude.
// No barriers and no cross-lane traffic: the block may have any size up to 64.  The recurrences are bound by memory
// latency (a wave's loads per step are all it has in flight), so small n is spread over MORE waves with fewer lanes each
// (invit_block: n = 1024 -> 128 waves of 8 lanes): four to eight times the loads in flight.
__global__ __launch_bounds__(64) void invit_kernel(int n, const double *__restrict__ d, const double *__restrict__ e,
                                                   const double *__restrict__ last_diag, const double *__restrict__ sigma,
                                                   const double *__restrict__ scal, double *__restrict__ F_ra,
                                                   double *__restrict__ F_b, double *__restrict__ F_d2, double *__restrict__ F_c,
                                                   unsigned char *__restrict__ F_sw, double *__restrict__ Y, int iterations, int refine)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const double tn = scal[0];
    const size_t ld = (size_t)n;
    if (!(tn > 0.0) || n == 1) { // the zero matrix (or 1 x 1): the unit vectors
        for (int k = 0; k < n; ++k) Y[(size_t)k * ld + v] = (k == v) ? 1.0 : 0.0;
        return;
    }
    const double s = scalbn(1.0, -ilogb(tn) - 1), sg = sigma[v] * s, tol = EV_EPS;
    const double dlast = last_diag[0];
    // Every sweep below is a recurrence over k whose loads do not depend on it: they are issued IVB steps at a time
    // (one memory round trip per block instead of one per step -- a wave has nothing else to hide the latency with),
    // unconditionally (full blocks, then a step-by-step tail), with running pointers.
    constexpr int IVB = 16;
    const size_t bstep = (size_t)IVB * ld;
    // ---- factorisation (kept for a later call with refine = 1: one more solve from the vectors as they are)
    if (!refine) {
        double ak = d[0] * s - sg, bk = e[0] * s;
        auto step = [&](int k, double ec, double dn, double en) {
            const double ck = ec * s, an = dn * s - sg, bn = en * s;
            const bool swap = fabs(ck) > fabs(ak);
            double piv, mult, bo, d2o, anew, bnew;
            if (swap) { // row k+1 = (ck, an, bn) becomes the pivot row
                mult = ak / ck;
                piv = ck;
                bo = an;
                d2o = bn;
                anew = bk - mult * an;
                bnew = -(mult * bn);
            } else {
                mult = (ak == 0.0) ? 0.0 : ck / ak; // (ak = 0 means ck = 0 as well: nothing to eliminate)
                piv = ak;
                bo = bk;
                d2o = 0.0;
                anew = an - mult * bk;
                bnew = bn;
            }
            if (fabs(piv) < tol) piv = (piv < 0.0) ? -tol : tol;
            const size_t at = (size_t)k * ld + v;
            F_ra[at] = 1.0 / piv;
            F_b[at] = bo;
            F_d2[at] = d2o;
            F_c[at] = mult;
            F_sw[at] = swap ? 1 : 0;
            ak = anew;
            bk = bnew;
        };
        int k = 0;
        for (; k + IVB + 1 < n; k += IVB) { // steps k .. k+IVB-1 read d[k+1 .. k+IVB], e[k .. k+IVB]: all below n-1
            double dn[IVB], ee[IVB + 1];
#pragma unroll
            for (int u = 0; u < IVB; ++u) dn[u] = d[k + 1 + u];
#pragma unroll
            for (int u = 0; u <= IVB; ++u) ee[u] = e[k + u];
#pragma unroll
            for (int u = 0; u < IVB; ++u) step(k + u, ee[u], dn[u], ee[u + 1]);
        }
        for (; k < n - 1; ++k) step(k, e[k], (k + 1 == n - 1) ? dlast : d[k + 1], (k + 2 < n) ? e[k + 1] : 0.0);
        double piv = ak;
        if (fabs(piv) < tol) piv = (piv < 0.0) ? -tol : tol;
        F_ra[(size_t)(n - 1) * ld + v] = 1.0 / piv;
        F_b[(size_t)(n - 1) * ld + v] = 0.0; // (row n-1 has no superdiagonals: the solve reads these unconditionally)
        F_d2[(size_t)(n - 1) * ld + v] = 0.0;
    }
    // ---- inverse iteration
    double scale = 1.0, ss = 0.0;
    for (int it = 0; it < iterations; ++it) {
        // forward: y <- L^{-1} P y  (rows k, k+1 -> row k final, carry)
        const bool fresh = (it == 0) && !refine;
        double yk = fresh ? invit_start((unsigned)v, 0u) : Y[v] * scale;
        {
            const double *pc = F_c + v;
            const unsigned char *ps = F_sw + v;
            double *py = Y + v;
            auto fstep = [&](double yn, double c, unsigned char swb, double *dst) {
                const bool w = swb != 0;
                const double out = w ? yn : yk, carry = w ? yk - c * yn : yn - c * yk;
                *dst = out;
                yk = carry;
            };
            int k = 0;
            if (fresh) {
                for (; k < n - 1; ++k) fstep(invit_start((unsigned)v, (unsigned)(k + 1)), pc[(size_t)k * ld], ps[(size_t)k * ld], py + (size_t)k * ld);
            } else {
                for (; k + IVB <= n - 1; k += IVB, pc += bstep, ps += bstep, py += bstep) {
                    double yn[IVB], c[IVB];
                    unsigned char sw[IVB];
#pragma unroll
                    for (int u = 0; u < IVB; ++u) {
                        yn[u] = py[(size_t)(u + 1) * ld];
                        c[u] = pc[(size_t)u * ld];
                        sw[u] = ps[(size_t)u * ld];
                    }
#pragma unroll
                    for (int u = 0; u < IVB; ++u) fstep(yn[u] * scale, c[u], sw[u], py + (size_t)u * ld);
                }
                for (int u = 0; k < n - 1; ++k, ++u) fstep(py[(size_t)(u + 1) * ld] * scale, pc[(size_t)u * ld], ps[(size_t)u * ld], py + (size_t)u * ld);
            }
            Y[(size_t)(n - 1) * ld + v] = yk;
        }
        // backward: y <- U^{-1} y
        double y1 = 0.0, y2 = 0.0, mx = 0.0;
        ss = 0.0;
        {
            auto bstepf = [&](double t, double fb, double fd, double ra, double *dst) {
                const double x = (t - fb * y1 - fd * y2) * ra;
                *dst = x;
                y2 = y1;
                y1 = x;
                mx = fmax(mx, fabs(x));
                ss += x * x;
            };
            int k = n - 1;
            for (; k - IVB + 1 >= 0; k -= IVB) {
                const size_t top = (size_t)k * ld + v;
                double t[IVB], fb[IVB], fd[IVB], ra[IVB];
#pragma unroll
                for (int u = 0; u < IVB; ++u) {
                    const size_t at = top - (size_t)u * ld;
                    t[u] = Y[at];
                    ra[u] = F_ra[at];
                    fb[u] = F_b[at];
                    fd[u] = F_d2[at];
                }
#pragma unroll
                for (int u = 0; u < IVB; ++u) bstepf(t[u], fb[u], fd[u], ra[u], Y + (top - (size_t)u * ld));
            }
            for (; k >= 0; --k) {
                const size_t at = (size_t)k * ld + v;
                bstepf(Y[at], F_b[at], F_d2[at], F_ra[at], Y + at);
            }
        }
        scale = 1.0 / mx;
    }
    const double rn = 1.0 / sqrt(ss);
#pragma unroll 8
    for (int k = 0; k < n; ++k) {
        const size_t at = (size_t)k * ld + v;
        Y[at] = Y[at] * rn;
    }
}

// out[0] = max over the lower triangle of |G - I|; frob2 (may be NULL) += the squared Frobenius norm of G - I (both
// triangles).  grid n (a column per workgroup), block 256.
__global__ __launch_bounds__(256) void gram_offdiag_kernel(int n, const double *G, int ld, double *out, double *frob2)
{
    __shared__ double red[2][4];
    const int j = blockIdx.x;
    double mx = 0.0, sq = 0.0;
    bool bad = false;
    for (int i = j + threadIdx.x; i < n; i += 256) {
        const double g = G[(size_t)j * ld + i] - ((i == j) ? 1.0 : 0.0);
        bad = bad || !(g == g);
        mx = fmax(mx, fabs(g));
        sq += (i == j) ? g * g : 2.0 * (g * g);
    }
    if (bad) mx = __builtin_inf();
    for (int o = 32; o > 0; o >>= 1) {
        mx = fmax(mx, __shfl_xor(mx, o));
        sq += __shfl_xor(sq, o);
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = mx;
        red[1][threadIdx.x >> 6] = sq;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomic_max_nonneg(out, fmax(fmax(red[0][0], red[0][1]), fmax(red[0][2], red[0][3])));
        if (frob2) atomicAdd(frob2, (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
    }
}

// G <- 3/2 I - 1/2 G (the Newton-Schulz multiplier).  grid (ceil(n / 256), n), block 256.
__global__ __launch_bounds__(256) void newton_schulz_matrix_kernel(int n, double *G, int ld)
{
    const int i = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;
    if (i < n) G[(size_t)j * ld + i] = ((i == j) ? 1.5 : 0.0) - 0.5 * G[(size_t)j * ld + i];
}

// out[0] = max_v ||T y_v - w_v y_v||_inf (T = the tridiagonal d, e; y_v = row v of Y).  One lane per vector, any block
// size up to 64 (see invit_kernel).
__global__ __launch_bounds__(64) void tri_residual_kernel(int n, const double *__restrict__ d, const double *__restrict__ e,
                                                          const double *__restrict__ last_diag, const double *__restrict__ w,
                                                          const double *__restrict__ Y, double *out)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const size_t ld = (size_t)n;
    const double lam = w[v];
    double mx = 0.0, ym = 0.0, yk = Y[v];
    bool bad = false;
    constexpr int B = 16;
    for (int k0 = 0; k0 < n; k0 += B) {
        double yn[B], dk[B], ek[B];
#pragma unroll
        for (int u = 0; u < B; ++u) {
            const int k = (k0 + u < n) ? k0 + u : n - 1;
            yn[u] = (k + 1 < n) ? Y[(size_t)(k + 1) * ld + v] : 0.0;
            dk[u] = (k == n - 1) ? last_diag[0] : d[k];
            ek[u] = (k + 1 < n) ? e[k] : 0.0;
        }
        double em = (k0 > 0) ? e[k0 - 1] : 0.0;
#pragma unroll
        for (int u = 0; u < B; ++u) {
            if (k0 + u < n) {
                const double r = (dk[u] - lam) * yk + em * ym + ek[u] * yn[u];
                bad = bad || !(r == r);
                mx = fmax(mx, fabs(r));
                ym = yk;
                yk = yn[u];
                em = ek[u];
            }
        }
    }
    if (bad) mx = __builtin_inf();
    atomic_max_nonneg(out, mx);
}

// Z(:, j) = H_0 H_1 ... H_{n-3} y_j, y_j = row j of Y.  NC eigenvectors live in the registers of NWV waves (component
// tid + 64 NWV i of vector blockIdx.x NC + c in z[c][i]); reflector k (column k of V below the subdiagonal, stored as
// u = sqrt(tau) v: nothing but the column is read -- a scalar load per reflector cannot be waited for without
// waiting for the newest one too) acts on components k+1 .. n-1 and is applied as z -= (u.z) u.  One wave per SIMD and nothing to
// overlap with: the kernel is bound by the instructions it issues per reflector, so the reflectors are taken in groups
// that start in the same register block IB = (k+1) / (64 NWV) -- blocks below IB are not touched, decided at compile time,
// and nothing is masked: the reflectors live in a zero-padded matrix of their own (ld = 64 NWV R); the wave sum is the
// DPP / permlane one of fl_reduce.hpp; the next NB - 1 reflectors (and their tau) are in flight while one is applied.
template <int NWV, int R, int NC, int NB, int IB> struct BackBlock {
    static constexpr int NT = 64 * NWV;
    // No branch and no mask around the loads, here or in the loop that calls this: a conditionally issued (or selected:
    // the compiler sinks the load into the select) load makes every wait a vmcnt(0), which waits for the newest prefetch
    // too.  V is zero above the reflector and in the padding rows n .. ldv-1 (ldv >= NT R).
    static __device__ __forceinline__ void load(int k, int klo, int tid, const double *__restrict__ V, int ldv, double (&vv)[R])
    {
        k = (k < klo) ? klo : k; // below the group: a valid column again, never applied
        const double *col = V + (size_t)k * ldv + tid;
#pragma unroll
        for (int i = IB; i < R; ++i) vv[i] = col[NT * i];
    }
    static __device__ __forceinline__ void apply(int k, int klo, int tid, const double (&vv)[R], double (&z)[NC][R],
                                                 double (*part)[NC][NWV > 1 ? NWV : 1], int &flip)
    {
        double p[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            double p0 = 0.0, p1 = 0.0;
#pragma unroll
            for (int i = IB; i < R; ++i) {
                if ((i - IB) & 1) p1 = __builtin_fma(vv[i], z[c][i], p1);
                else p0 = __builtin_fma(vv[i], z[c][i], p0);
            }
            p[c] = wave_allreduce(p0 + p1);
        }
        if constexpr (NWV > 1) {
            if ((tid & 63) == 0)
#pragma unroll
                for (int c = 0; c < NC; ++c) part[flip][c][tid >> 6] = p[c];
            __syncthreads();
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                p[c] = 0.0;
#pragma unroll
                for (int q = 0; q < NWV; ++q) p[c] += part[flip][c][q];
            }
            flip ^= 1; // (the slot written two reflectors later: everybody has passed the barrier in between)
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const double sc = -p[c];
#pragma unroll
            for (int i = IB; i < R; ++i) z[c][i] = __builtin_fma(sc, vv[i], z[c][i]);
        }
    }
    static __device__ __forceinline__ void run(int n, int tid, const double *__restrict__ V, int ldv, double (&z)[NC][R],
                                               double (*part)[NC][NWV > 1 ? NWV : 1], int &flip)
    {
        // reflectors with NT IB <= k + 1 < NT (IB + 1), the last one first
        const int khi = (n - 3 < NT * IB + NT - 2) ? n - 3 : NT * IB + NT - 2, klo = (NT * IB - 1 > 0) ? NT * IB - 1 : 0;
        if (khi >= klo) {
            double vbuf[NB][R];
#pragma unroll
            for (int b = 0; b < NB - 1; ++b) load(khi - b, klo, tid, V, ldv, vbuf[b]);
            int k = khi;
            for (; k - (NB - 1) >= klo; k -= NB) { // full groups: straight-line
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    load(k - b - (NB - 1), klo, tid, V, ldv, vbuf[(b + NB - 1) % NB]);
                    apply(k - b, klo, tid, vbuf[b], z, part, flip);
                }
            }
#pragma unroll
            for (int b = 0; b < NB - 1; ++b) // the rest (fewer than NB): already in the ring
                if (k - b >= klo) apply(k - b, klo, tid, vbuf[b], z, part, flip);
        }
        if constexpr (IB > 0) BackBlock<NWV, R, NC, NB, IB - 1>::run(n, tid, V, ldv, z, part, flip);
    }
};

// grid ceil(n / NC), block 64 NWV
template <int NWV, int R, int NC, int NB>
__global__ __launch_bounds__(64 * NWV) void backtransform_kernel(int n, const double *__restrict__ Y,
                                                                const double *__restrict__ V, int ldv, double *__restrict__ Z, int ldz)
{
    constexpr int NT = 64 * NWV;
    __shared__ double part[2][NC][NWV > 1 ? NWV : 1];
    const int j0 = blockIdx.x * NC, tid = threadIdx.x;
    double z[NC][R];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int comp = tid + NT * i;
            z[c][i] = (comp < n && j0 + c < n) ? Y[(size_t)comp * n + j0 + c] : 0.0;
        }
    int flip = 0;
    BackBlock<NWV, R, NC, NB, R - 1>::run(n, tid, V, ldv, z, part, flip);
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int comp = tid + NT * i;
            if (comp < n && j0 + c < n) Z[(size_t)(j0 + c) * ldz + comp] = z[c][i];
        }
}

// lanes per wave of the one-lane-per-vector kernels: at least ~128 waves
inline int invit_block(int n) { return n >= 8192 ? 64 : n >= 4096 ? 32 : n >= 2048 ? 16 : 8; }

} // namespace fl

extern "C" {

// leading dimension of the kept reflectors: the register blocks of backtransform_kernel's geometry for this n
static int reflector_ld(int n) { return n <= 2048 ? (n + 255) / 256 * 256 : (n + 1023) / 1024 * 1024; }

// doubles: tridiagonalisation front (6 n + 4), tau, sigma (n each), 8 scalars, Y, G (n^2 each), the factorisation's
// F_ra, F_b, F_d2, F_c (n^2 each), F_sw (n^2 bytes), the reflectors (n x reflector_ld), the blocked Cholesky's scratch
size_t fl_dsyev_vectors_workspace_bytes(int n)
{
    if (n <= 0) return 0;
    const size_t nn = (size_t)n * n;
    return (8 * (size_t)n + 12 + 6 * nn + (nn + 7) / 8 + (size_t)n * reflector_ld(n)) * sizeof(double) +
           fl_chol_blocked_workspace_bytes(1, n, n);
}

// A_dev (n x n column-major, lda = n, lower triangle referenced) -> its normalised eigenvectors (columns), w_dev: the
// eigenvalues in ascending order.  Returns FL_OK; 1 when the device-side check of the basis failed (A_dev and w_dev are
// then undefined: the caller falls back to fl_dsyev_jacobi on a fresh copy of A); a negative FL_ERR_* code otherwise.
// quality_host (may be NULL): [0] = max |Y Y^T - I| of the tridiagonal's vectors, [1] = max |T y - lambda y| / ||T||,
// [2] = Cholesky-QR passes used.  Synchronises the stream (the check decides on the host).
int fl_dsyev_vectors(int n, double *A_dev, int lda, double *w_dev, void *workspace_dev, size_t workspace_bytes, double *quality_host,
                     void *stream)
{
    if (!A_dev || !w_dev || n <= 0 || lda != n) return FL_ERR_INVALID_ARGUMENT;
    if (n > 6144) return FL_ERR_UNSUPPORTED_SIZE;
    if (!workspace_dev || workspace_bytes < fl_dsyev_vectors_workspace_bytes(n)) return FL_ERR_WORKSPACE;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return FL_ERR_NO_DEVICE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t nn = (size_t)n * n;
    double *front = static_cast<double *>(workspace_dev);
    double *tauv = front + 6 * (size_t)n + 4, *sigma = tauv + n, *scal = sigma + n;
    double *Y = scal + 8, *G = Y + nn, *F_ra = G + nn, *F_b = F_ra + nn, *F_d2 = F_b + nn, *F_c = F_d2 + nn;
    unsigned char *F_sw = reinterpret_cast<unsigned char *>(F_c + nn);
    const int ldv = reflector_ld(n);
    double *Vk = F_c + nn + (nn + 7) / 8, *cws = Vk + (size_t)n * ldv;
    const size_t cwsb = fl_chol_blocked_workspace_bytes(1, n, n);
    double *dvec = front + 4 * (size_t)n + 4, *evec = dvec + n, *last = A_dev + (size_t)(n - 1) * n + (n - 1);
    int32_t *info = reinterpret_cast<int32_t *>(scal + 4);
    if (hipMemsetAsync(tauv, 0, sizeof(double) * (2 * (size_t)n + 8), st) != hipSuccess) return FL_ERR_LAUNCH;
    const char *dbg = std::getenv("FL_DSYEV_DEBUG");
    const bool debug = dbg && dbg[0] == '1';
    auto now = [&] {
        if (debug) (void)hipStreamSynchronize(st);
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
    };
    const double t0 = now();
    if (hipMemsetAsync(Vk, 0, sizeof(double) * (size_t)n * ldv, st) != hipSuccess) return FL_ERR_LAUNCH;
    int rc = fl_sytrd_values(n, A_dev, w_dev, front, tauv, Vk, ldv, stream);
    if (rc != FL_OK) return rc;
    const double t1 = now();
    const size_t lds_shift = (size_t)2 * n * sizeof(double);
    if (lds_shift > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(fl::invit_shift_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_shift) != hipSuccess)
        return FL_ERR_LAUNCH;
    hipLaunchKernelGGL(fl::invit_shift_kernel, dim3(1), dim3(256), lds_shift, st, n, w_dev, dvec, evec, last, sigma, scal);
    const int ivb = fl::invit_block(n);
    hipLaunchKernelGGL(fl::invit_kernel, dim3((n + ivb - 1) / ivb), dim3(ivb), 0, st, n, dvec, evec, last, sigma, scal, F_ra, F_b, F_d2,
                       F_c, F_sw, Y, 3, 0);
    if (fl::launch_status() != FL_OK) return FL_ERR_LAUNCH;
    // Orthonormal rows.  G = Y Y^T decides how:
    //   ||G - I||_F < 1e-8 (separated eigenvalues: the usual case): ONE Newton-Schulz step Y <- (3/2 I - 1/2 G) Y, a single
    //     product on the matrix cores; what is left of G - I is 3/8 ||G - I||^2 < eps;
    //   otherwise Cholesky-QR: G = L L^T (blocked Cholesky), Y <- L^-1 Y;
    //   ||G - I||_F >= 1/2 (clusters, multiple eigenvalues: their vectors came out as independent but ill-conditioned
    //     combinations, and whitening those amplifies the rounding noise OUTSIDE the cluster's invariant subspace by their
    //     condition number): one more solve from the whitened vectors puts that noise back to eps, then Cholesky-QR
    //     again, repeated (without further solves) until the Gram matrix it started from was close to the identity.
    double h[6] = {0, 0, 0, 0, 0, 0};
    const double t2 = now();
    int passes = 0;
    for (int pass = 0; pass < 5; ++pass) {
        if (pass == 1) {
            hipLaunchKernelGGL(fl::invit_kernel, dim3((n + ivb - 1) / ivb), dim3(ivb), 0, st, n, dvec, evec, last, sigma, scal, F_ra,
                               F_b, F_d2, F_c, F_sw, Y, 1, 1);
            if (fl::launch_status() != FL_OK) return FL_ERR_LAUNCH;
        }
        if (hipMemsetAsync(scal + 1, 0, sizeof(double), st) != hipSuccess || hipMemsetAsync(scal + 5, 0, sizeof(double), st) != hipSuccess)
            return FL_ERR_LAUNCH;
        rc = fl_dgemm_strided(0, 1, n, n, n, 1.0, Y, n, 0, Y, n, 0, 0.0, G, n, 0, 1, 0, st); // (both triangles: Newton-Schulz wants them)
        if (rc != FL_OK) return rc;
        hipLaunchKernelGGL(fl::gram_offdiag_kernel, dim3(n), dim3(256), 0, st, n, G, n, scal + 1, scal + 5);
        if (hipMemcpyAsync(h, scal, sizeof(double) * 6, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
            return FL_ERR_LAUNCH;
        ++passes;
        const double fro = std::sqrt(h[5]);
        if (debug) std::fprintf(stderr, "fl_dsyev_vectors n=%d pass %d: max|G-I| %.3e  ||G-I||_F %.3e\n", n, pass, h[1], fro);
        if (!(h[1] == h[1]) || !(fro == fro) || std::isinf(fro)) return 1;
        if (fro < 1e-8) {
            hipLaunchKernelGGL(fl::newton_schulz_matrix_kernel, dim3((n + 255) / 256, n), dim3(256), 0, st, n, G, n);
            rc = fl_dgemm_strided(0, 0, n, n, n, 1.0, G, n, 0, Y, n, 0, 0.0, F_ra, n, 0, 1, 0, st); // (the factorisation is not needed any more)
            if (rc != FL_OK) return rc;
            Y = F_ra;
            break;
        }
        rc = fl_chol_whiten(n, n, G, n, Y, info, cws, cwsb, st);
        if (rc != FL_OK) return rc;
        int32_t hinfo = 0;
        if (hipMemcpyAsync(&hinfo, info, sizeof hinfo, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
            return FL_ERR_LAUNCH;
        if (hinfo != 0) return 1; // vectors dependent to rounding: not this path's case
        if (fro < 0.5) break;     // cond(Y)^2 <= (1 + delta) / (1 - delta), delta = ||G - I||_2 <= ||G - I||_F < 1/2
    }
    const double t3 = now();
    // the check: orthogonality of the rows of Y and the tridiagonal residuals
    if (hipMemsetAsync(scal + 2, 0, 2 * sizeof(double), st) != hipSuccess) return FL_ERR_LAUNCH;
    rc = fl_dgemm_strided(0, 1, n, n, n, 1.0, Y, n, 0, Y, n, 0, 0.0, G, n, 0, 1, 1, st);
    if (rc != FL_OK) return rc;
    hipLaunchKernelGGL(fl::gram_offdiag_kernel, dim3(n), dim3(256), 0, st, n, G, n, scal + 2, static_cast<double *>(nullptr));
    hipLaunchKernelGGL(fl::tri_residual_kernel, dim3((n + ivb - 1) / ivb), dim3(ivb), 0, st, n, dvec, evec, last, w_dev, Y, scal + 3);
    if (hipMemcpyAsync(h, scal, sizeof(double) * 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        return FL_ERR_LAUNCH;
    const double tn = h[0], orth = h[2], res = (tn > 0.0) ? h[3] / tn : h[3];
    if (debug) std::fprintf(stderr, "fl_dsyev_vectors n=%d check: orth %.3e  residual/||T|| %.3e  (||T|| %.3e)\n", n, orth, res, tn);
    if (quality_host) {
        quality_host[0] = orth;
        quality_host[1] = res;
        quality_host[2] = passes;
    }
    const double t4 = now();
    if (!(orth <= 512.0 * fl::EV_EPS) || !(res <= 512.0 * fl::EV_EPS)) return 1;
    // back-transformation into G's place, then over A
    double *Z = G;
#define FL_BT(NWV, R, NC, NB)                                                                                                  \
    hipLaunchKernelGGL((fl::backtransform_kernel<NWV, R, NC, NB>), dim3((n + NC - 1) / NC), dim3(64 * NWV), 0, st, n, Y, Vk, ldv, Z, n)
    // register blocks per vector, in steps of four: n <= 64 NWV R, n > 64 NWV (R - 4)
    if (n <= 256) FL_BT(1, 4, 2, 4);
    else if (n <= 512) FL_BT(1, 8, 2, 4);
    else if (n <= 768) FL_BT(1, 12, 2, 4);
    else if (n <= 1024) FL_BT(1, 16, 2, 4); // (two vectors per wave share each reflector's loads: 0.34 -> 0.28 ms at n = 1000)
    else if (n <= 1280) FL_BT(1, 20, 1, 2);
    else if (n <= 1536) FL_BT(1, 24, 1, 2);
    else if (n <= 1792) FL_BT(1, 28, 1, 2);
    else if (n <= 2048) FL_BT(1, 32, 1, 2);
    else if (n <= 3072) FL_BT(4, 12, 2, 4); // (beyond 2048: four waves per vector and TWO vectors per workgroup -- there the
    else if (n <= 4096) FL_BT(4, 16, 2, 4); //  reflectors' L2 traffic binds: n = 4096: 18.3 -> 10.6 ms)
    else if (n <= 5120) FL_BT(4, 20, 2, 2);
    else FL_BT(4, 24, 2, 2);
#undef FL_BT
    if (fl::launch_status() != FL_OK) return FL_ERR_LAUNCH;
    if (hipMemcpyAsync(A_dev, Z, sizeof(double) * nn, hipMemcpyDeviceToDevice, st) != hipSuccess) return FL_ERR_LAUNCH;
    if (debug)
        std::fprintf(stderr, "fl_dsyev_vectors n=%d: tridiagonalisation + eigenvalues %.2f ms, inverse iteration %.2f, orthonormalisation %.2f, check %.2f, back-transformation %.2f\n",
                     n, t1 - t0, t2 - t1, t3 - t2, t4 - t3, now() - t4);
    return FL_OK;
}

} // extern "C"
