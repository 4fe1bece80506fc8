// fl_linesearch.hpp -- the reference's four line searchers as ONE resumable
// state machine (host + device).
//
// Reference: /root/reference/source/NonlinearOptimization.f90
//   Wolfe 1286-1371 (= Wolfe_fdwithf 1373-1459), StrongWolfe 1462-1580,
//   StrongWolfe_fdwithf 1582-1698, and their internal zoom procedures.
//
// The reference walks these as nested loops around user callbacks.  On the GPU a
// whole workgroup owns one problem and every scalar below is uniform across the
// workgroup, so the search is restated as a machine that is stepped with the
// result of the last evaluation and answers with the next evaluation request:
//
//     int rq = ls.begin(a0, fx, phid0);
//     while (rq) { evaluate at x0 + ls.a_eval * p what rq asks for;  rq = ls.step(f, g.p); }
//     // finished: step length ls.a, objective ls.fx, x and g are those of the last evaluation
//
// Request bits: FL_REQ_F (objective), FL_REQ_G (gradient; the caller returns
// g.p), FL_REQ_SAME (the point is the one of the previous request: nothing to
// move).  The bits also say how the reference would have counted f / f'
// callback invocations.  Floating-point expressions keep the Fortran source's
// evaluation order (no reassociation, no FMA) -- including its quirks: the
// "search for larger a" branch of StrongWolfe calls zoom and then keeps looping
// with fx = fx0 (NO.f90:1507-1514) while the _fdwithf twin returns
// (NO.f90:1628-1632).
#pragma once
#ifndef __HIPCC_RTC__
#include <math.h>
#endif

#if defined(__HIPCC__)
#define FL_HD __host__ __device__ __forceinline__
#else
#define FL_HD inline
#endif

#define FL_REQ_F 1
#define FL_REQ_G 2
#define FL_REQ_SAME 4

// A zoom that works needs ~60 trials to halve an interval down to 1e-15; the reference's has no iteration limit at all
// (NO.f90:1557-1579, Wolfe's: 1347-1370) and can cycle between two points for ever: the machine ends the
// machine counts its zoom trials and the fused kernels end a problem beyond this many (Solver::must_stop: FL_STATUS_STALLED).
#ifndef FL_ZOOM_CAP
#define FL_ZOOM_CAP 65536
#endif

namespace fl {

struct LineSearch {
    enum State {
        DONE = 0,
        SW_FIRST, SW_FIRST_G, SW_SHRINK_A, SW_GROW, SW_V_F, SW_V_G, SW_V_GOLD, SW_V_SHRINK, SW_LAST_G, SW_ZOOM,
        W_FIRST, W_GROW, W_GROW_G, W_SHRINK, W_SHRINK_G, W_LAST_G, WZ_F, WZ_G
    };
    // search parameters
    double c1, c2abs, incr, fx0, phid0;
    int fused; // the *_fdwithf variants (caller passed f_fd)
    // running values (names follow the Fortran)
    double a, aold, fx, fold, phidnew, phidold;
    double low, up, flow, fup, phidlow, phidup, plma; // zoom's arguments; plma = phidlow_m_a
    int st, zret;
    int zn; // trials of the zoom in this search (more than FL_ZOOM_CAP: stalled() -- the reference's zoom has no limit)
    double a_eval; // where the pending request is to be evaluated

    // Every value of the machine is uniform across the workgroup.  On the GPU, pin the state to
    // scalar registers after each step (v_readfirstlane) so that it does not occupy one vector
    // register pair per value in every lane; a no-op on the host.
    FL_HD static double uni(double v)
    {
#if defined(__HIP_DEVICE_COMPILE__)
        const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
        const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
        return __hiloint2double(hi, lo);
#else
        return v;
#endif
    }
#ifndef FL_UNI_LEVEL
#define FL_UNI_LEVEL 0
#endif
    template <int LEVEL = FL_UNI_LEVEL> FL_HD void uniformize()
    {
        // Measured on the north-star workload (profiles/r01/geometry_ab.txt): pinning the whole
        // machine pushes the kernel over the SGPR budget and the spills (v_readlane) cost more
        // than the occupancy gains: level 0 = 25.8, 1 = 23.4, 2 = 23.0 M it/s.  The dense kernels at 8
        // elements per thread use level 2: there the vector registers are what is short.
        if constexpr (LEVEL >= 1) { // the values every trial touches
            c2abs = uni(c2abs); fx0 = uni(fx0); phid0 = uni(phid0);
            a = uni(a); aold = uni(aold); fx = uni(fx); fold = uni(fold); phidnew = uni(phidnew); phidold = uni(phidold);
            a_eval = uni(a_eval);
        }
        if constexpr (LEVEL >= 2) { // zoom's bracket and the constants
            c1 = uni(c1); incr = uni(incr);
            low = uni(low); up = uni(up); flow = uni(flow); fup = uni(fup); phidlow = uni(phidlow); phidup = uni(phidup);
            plma = uni(plma);
        }
#if defined(__HIP_DEVICE_COMPILE__)
        st = __builtin_amdgcn_readfirstlane(st);
        zret = __builtin_amdgcn_readfirstlane(zret);
        fused = __builtin_amdgcn_readfirstlane(fused);
#endif
    }

    FL_HD static double dmax(double u, double v) { return u > v ? u : v; }
    FL_HD static double dmin(double u, double v) { return u < v ? u : v; }

    FL_HD int req(int kind, double at, int same)
    {
        a_eval = at;
        return kind | (same ? FL_REQ_SAME : 0);
    }
    FL_HD int done()
    {
        st = DONE;
        return 0;
    }
    FL_HD bool stalled() const { return zn > FL_ZOOM_CAP; } // (until the next begin())
    FL_HD bool armijo() const { return fx <= fx0 + c1 * a * phid0; }

    // c1, c2 as clamped by the solver; increment as passed (fail-safe NO.f90:1478)
    FL_HD int begin(int strong, int fused_, double c1_, double c2_, double increment, double a0, double fx_in,
                    double phid0_in)
    {
        zn = 0;
        c1 = c1_;
        incr = dmax(1.0 + 1e-15, increment);
        fused = fused_;
        fx0 = fx_in;
        fx = fx_in;
        phid0 = phid0_in;
        c2abs = c2_ * fabs(phid0_in);
        a = a0;
        zret = 0;
        aold = fold = phidnew = phidold = 0.0;
        low = up = flow = fup = phidlow = phidup = plma = 0.0;
        if (strong) {
            st = SW_FIRST;
            return req(fused ? (FL_REQ_F | FL_REQ_G) : FL_REQ_F, a, 0);
        }
        st = W_FIRST; // Wolfe_fdwithf never uses f_fd (NO.f90:1373)
        return req(FL_REQ_F, a, 0);
    }

    // ---- strong Wolfe helpers
    FL_HD int sw_shrink_next(int state) // NO.f90:1489-1490 / 1532-1533
    {
        aold = a;
        fold = fx;
        phidold = phidnew;
        a = aold / incr;
        st = state;
        return req(FL_REQ_F | FL_REQ_G, a, 0);
    }
    FL_HD int sw_grow_next() // NO.f90:1500-1501
    {
        aold = a;
        fold = fx;
        phidold = phidnew;
        a = aold * incr;
        st = SW_GROW;
        return req(FL_REQ_F | FL_REQ_G, a, 0);
    }
    FL_HD int sw_v_next() // NO.f90:1519-1520
    {
        aold = a;
        fold = fx;
        a = aold / incr;
        st = SW_V_F;
        return req(FL_REQ_F, a, 0);
    }
    FL_HD int sw_first_slope() // NO.f90:1486-1516 entry
    {
        if (phidnew > 0.0) {
            if (fabs(phidnew) <= c2abs) return done();
            return sw_shrink_next(SW_SHRINK_A);
        }
        return sw_grow_next();
    }
    FL_HD void zoom_args(double l, double u, double fl_, double fu, double pl, double pu, int ret)
    {
        low = l;
        up = u;
        flow = fl_;
        fup = fu;
        phidlow = pl;
        phidup = pu;
        zret = ret;
    }
    FL_HD int sw_zoom_next() // cubic interpolation, NO.f90:1562-1567
    {
        double d1 = phidlow + phidup - 3.0 * (flow - fup) / (low - up);
        double d2 = up - low;
        if (d2 > 0.0)
            d2 = sqrt(d1 * d1 - phidlow * phidup);
        else
            d2 = -sqrt(d1 * d1 - phidlow * phidup);
        a = up - (up - low) * (phidup + d2 - d1) / (phidup - phidlow + 2.0 * d2);
        if (!(a > dmin(low, up) && a < dmax(low, up))) a = (low + up) / 2.0;
        st = SW_ZOOM;
        return req(FL_REQ_F | FL_REQ_G, a, 0);
    }
    FL_HD int sw_zoom_ret()
    {
        if (zret == 0) return done();
        // NO.f90:1511-1512: zoom(atemp,aold,ftemp,fold,phidnew,phidold); fx=fx0 -- and the loop goes on.
        // zoom's phidlow is the caller's phidnew (passed by reference).
        fx = fx0;
        phidnew = phidlow;
        return sw_grow_next();
    }

    // ---- Wolfe helpers
    FL_HD int w_grow_next() // NO.f90:1309-1310
    {
        aold = a;
        fold = fx;
        a = aold * incr;
        st = W_GROW;
        return req(FL_REQ_F, a, 0);
    }
    FL_HD int w_shrink_next() // NO.f90:1326-1327
    {
        aold = a;
        fold = fx;
        a = aold / incr;
        st = W_SHRINK;
        return req(FL_REQ_F, a, 0);
    }
    FL_HD int wz_next() // quadratic interpolation, NO.f90:1353-1355
    {
        a = plma * a / 2.0 / (flow + plma - fup);
        if (!(a > low && a < up)) a = (low + up) / 2.0;
        st = WZ_F;
        return req(FL_REQ_F, a, 0);
    }
    FL_HD bool w_collapsed() const
    {
        return up - low < 1e-15 || (up - low) / dmax(fabs(low), fabs(up)) < 1e-15;
    }

    // fv = f at the requested point (if FL_REQ_F was set), pv = g.p there (if FL_REQ_G)
    FL_HD int step(double fv, double pv)
    {
        switch (st) {
        // ------------------------------------------------ StrongWolfe
        case SW_FIRST: // NO.f90:1482-1483 / 1604-1605
            fx = fv;
            if (armijo()) {
                if (!fused) {
                    st = SW_FIRST_G;
                    return req(FL_REQ_G, a, 1);
                }
                phidnew = pv;
                return sw_first_slope();
            }
            return sw_v_next();
        case SW_FIRST_G:
            phidnew = pv;
            return sw_first_slope();
        case SW_SHRINK_A: // NO.f90:1488-1497
        case SW_V_SHRINK: // NO.f90:1531-1540
            fx = fv;
            phidnew = pv;
            if (fx >= fold || phidnew <= 0.0) {
                zoom_args(aold, a, fold, fx, phidold, phidnew, 0);
                return sw_zoom_next();
            }
            if (a < 1e-15) return done();
            return sw_shrink_next(st);
        case SW_GROW: // NO.f90:1499-1515 / 1620-1634
            fx = fv;
            phidnew = pv;
            if (fx > fx0 + c1 * a * phid0 || fx >= fold) {
                zoom_args(aold, a, fold, fx, phidold, phidnew, 0);
                return sw_zoom_next();
            }
            if (phidnew > 0.0) {
                if (fabs(phidnew) <= c2abs) return done();
                zoom_args(a, aold, fx, fold, phidnew, phidold, fused ? 0 : 1);
                return sw_zoom_next();
            }
            return sw_grow_next();
        case SW_V_F: // NO.f90:1518-1546
            fx = fv;
            if (armijo()) {
                st = SW_V_G;
                return req(FL_REQ_G, a, 1);
            }
            if (a < 1e-15) {
                st = SW_LAST_G;
                return req(FL_REQ_G, a, 1);
            }
            return sw_v_next();
        case SW_V_G: // NO.f90:1522-1530
            phidnew = pv;
            if (fabs(phidnew) <= c2abs) return done();
            if (phidnew < 0.0) {
                st = SW_V_GOLD;
                return req(FL_REQ_G, aold, 0);
            }
            return sw_shrink_next(SW_V_SHRINK);
        case SW_V_GOLD: // NO.f90:1526-1529
            phidold = pv;
            zoom_args(a, aold, fx, fold, phidnew, phidold, 0);
            return sw_zoom_next();
        case SW_LAST_G:
            return done();
        case SW_ZOOM: { // NO.f90:1567-1577
            ++zn;
            fx = fv;
            const double pn = pv;
            if (fx > fx0 + c1 * a * phid0 || fx >= flow) {
                up = a;
                fup = fx;
                phidup = pn;
            } else {
                if (fabs(pn) <= c2abs) return sw_zoom_ret();
                if (pn * (up - low) >= 0.0) {
                    up = low;
                    fup = flow;
                    phidup = phidlow;
                }
                low = a;
                flow = fx;
                phidlow = pn;
            }
            if (fabs(up - low) < 1e-15 || fabs(up - low) / dmax(fabs(low), fabs(up)) < 1e-15) return sw_zoom_ret();
            return sw_zoom_next();
        }
        // ------------------------------------------------ Wolfe
        case W_FIRST: // NO.f90:1306-1307
            fx = fv;
            if (armijo()) return w_grow_next();
            return w_shrink_next();
        case W_GROW: // NO.f90:1311-1313
            fx = fv;
            if (fx > fx0 + c1 * a * phid0) {
                st = W_GROW_G;
                return req(FL_REQ_G, aold, 0);
            }
            return w_grow_next();
        case W_GROW_G: // NO.f90:1314-1321
            if (pv > c2abs) {
                a = aold;
                fx = fold;
                return done();
            }
            zoom_args(aold, a, fold, fx, pv, 0.0, 0);
            plma = phidlow * a;
            return wz_next();
        case W_SHRINK: // NO.f90:1328-1339
            fx = fv;
            if (armijo()) {
                st = W_SHRINK_G;
                return req(FL_REQ_G, a, 1);
            }
            if (a < 1e-15) {
                st = W_LAST_G;
                return req(FL_REQ_G, a, 1);
            }
            return w_shrink_next();
        case W_SHRINK_G: // NO.f90:1330-1335
            if (pv < c2abs) {
                zoom_args(a, aold, fx, fold, pv, 0.0, 0);
                plma = phidlow * a;
                return wz_next();
            }
            return done();
        case W_LAST_G:
            return done();
        case WZ_F: // NO.f90:1355-1362
            ++zn;
            fx = fv;
            if (fx > fx0 + c1 * a * phid0) {
                up = a;
                if (w_collapsed()) {
                    st = W_LAST_G;
                    return req(FL_REQ_G, a, 1);
                }
                fup = fx;
                return wz_next();
            }
            st = WZ_G;
            return req(FL_REQ_G, a, 1);
        case WZ_G: // NO.f90:1363-1367
            if (pv > c2abs) return done();
            low = a;
            if (w_collapsed()) return done();
            flow = fx;
            phidlow = pv;
            plma = phidlow * a;
            return wz_next();
        default:
            return done();
        }
    }
};

} // namespace fl
