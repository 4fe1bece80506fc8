// fl_linalg.cpp -- the LinearAlgebra entry points the reference's C++ header binds next to the optimisers
// (cpp/FortranLibrary.hpp:48-63): My_dgemm / My_dgemm_T (LinearAlgebra.f90:182-196: plain dgemm calls) and My_dsyev
// (879-887: dsyev 'L').  The reference hands these to MKL; here they are handed to the vendor libraries of the ROCm
// stack -- rocBLAS dgemm, rocSOLVER dsyev -- on the GPU: host arrays in, host arrays out, like the reference.
// Plain library calls (no kernel of this repository is involved); same results to rounding, eigenvectors up to sign.
// The libraries are opened on first use (dlopen), so libFL.so itself does not depend on them.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>

#include <cstdio>
#include <mutex>

namespace {

struct Vendor {
    void *blas = nullptr, *solver = nullptr;
    rocblas_handle handle = nullptr;
    decltype(&rocblas_create_handle) create = nullptr;
    decltype(&rocblas_dgemm) dgemm = nullptr;
    decltype(&rocsolver_dsyev) dsyev = nullptr;
    bool ok = false;
};
Vendor &vendor()
{
    static Vendor v;
    static std::once_flag once;
    std::call_once(once, [] {
        v.blas = dlopen("librocblas.so", RTLD_NOW | RTLD_GLOBAL);
        v.solver = dlopen("librocsolver.so", RTLD_NOW | RTLD_GLOBAL);
        if (!v.blas || !v.solver) return;
        v.create = reinterpret_cast<decltype(v.create)>(dlsym(v.blas, "rocblas_create_handle"));
        v.dgemm = reinterpret_cast<decltype(v.dgemm)>(dlsym(v.blas, "rocblas_dgemm"));
        v.dsyev = reinterpret_cast<decltype(v.dsyev)>(dlsym(v.solver, "rocsolver_dsyev"));
        v.ok = v.create && v.dgemm && v.dsyev && v.create(&v.handle) == rocblas_status_success;
    });
    return v;
}

struct DeviceBuffer {
    void *p = nullptr;
    explicit DeviceBuffer(size_t bytes) { (void)hipMalloc(&p, bytes ? bytes : 8); }
    ~DeviceBuffer() { if (p) (void)hipFree(p); }
    template <class T> T *as() const { return static_cast<T *>(p); }
};

// C(M,N) = op(A) B, column-major; transA: A is K x M
void gemm(bool transA, const double *A, const double *B, double *C, int M, int K, int N)
{
    Vendor &v = vendor();
    int ndev = 0;
    if (!v.ok || hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        std::fprintf(stderr, "FortranLibrary(MI355X) My_dgemm: no HIP device or rocBLAS; C is unchanged\n");
        return;
    }
    const size_t a = sizeof(double) * (size_t)M * K, b = sizeof(double) * (size_t)K * N, c = sizeof(double) * (size_t)M * N;
    DeviceBuffer Ad(a), Bd(b), Cd(c);
    const double one = 1.0, zero = 0.0;
    bool ok = Ad.p && Bd.p && Cd.p && hipMemcpy(Ad.p, A, a, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(Bd.p, B, b, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && v.dgemm(v.handle, transA ? rocblas_operation_transpose : rocblas_operation_none, rocblas_operation_none, M, N,
                       K, &one, Ad.as<double>(), transA ? K : M, Bd.as<double>(), K, &zero, Cd.as<double>(), M) ==
                   rocblas_status_success;
    ok = ok && hipMemcpy(C, Cd.p, c, hipMemcpyDeviceToHost) == hipSuccess;
    if (!ok) std::fprintf(stderr, "FortranLibrary(MI355X) My_dgemm: device error; C is undefined\n");
}

} // namespace

extern "C" {

// subroutine My_dgemm(A,B,C,M,K,N): C = A . B (LinearAlgebra.f90:182-188)
void __linearalgebra_MOD_my_dgemm(const double *A, const double *B, double *C, const int *M, const int *K, const int *N)
{
    gemm(false, A, B, C, *M, *K, *N);
}
// subroutine My_dgemm_T(A,B,C,M,K,N): C = A^T . B, A is K x M (LinearAlgebra.f90:190-196; FortranLibrary.hpp:50)
void __linearalgebra_MOD_my_dgemm_t(const double *A, const double *B, double *C, const int *M, const int *K, const int *N)
{
    gemm(true, A, B, C, *M, *K, *N);
}
// subroutine My_dsyev(jobtype,A,eigval,N): eigenvalues ascending, A <- normalised eigenvectors for 'V'
// (LinearAlgebra.f90:879-887: dsyev(jobtype,'L',...); "A will be overwritten even for 'N' job")
void __linearalgebra_MOD_my_dsyev(const char *jobtype, double *A, double *eigval, const int *N, int /*len_jobtype*/)
{
    Vendor &v = vendor();
    const int n = *N;
    int ndev = 0;
    if (!v.ok || hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        std::fprintf(stderr, "FortranLibrary(MI355X) My_dsyev: no HIP device or rocSOLVER; A is unchanged\n");
        return;
    }
    const size_t a = sizeof(double) * (size_t)n * n;
    DeviceBuffer Ad(a), Dd(sizeof(double) * n), Ed(sizeof(double) * n), info(sizeof(rocblas_int));
    const bool vec = jobtype && (*jobtype == 'V' || *jobtype == 'v');
    bool ok = Ad.p && Dd.p && Ed.p && info.p && hipMemcpy(Ad.p, A, a, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && v.dsyev(v.handle, vec ? rocblas_evect_original : rocblas_evect_none, rocblas_fill_lower, n, Ad.as<double>(), n,
                       Dd.as<double>(), Ed.as<double>(), info.as<rocblas_int>()) == rocblas_status_success;
    ok = ok && hipMemcpy(eigval, Dd.p, sizeof(double) * n, hipMemcpyDeviceToHost) == hipSuccess &&
         hipMemcpy(A, Ad.p, a, hipMemcpyDeviceToHost) == hipSuccess;
    if (!ok) std::fprintf(stderr, "FortranLibrary(MI355X) My_dsyev: device error; A, eigval are undefined\n");
}
// the ifort manglings (FortranLibrary.hpp:27-43)
void linearalgebra_mp_my_dgemm_(const double *A, const double *B, double *C, const int *M, const int *K, const int *N)
{
    gemm(false, A, B, C, *M, *K, *N);
}
void linearalgebra_mp_my_dgemm_t_(const double *A, const double *B, double *C, const int *M, const int *K, const int *N)
{
    gemm(true, A, B, C, *M, *K, *N);
}
void linearalgebra_mp_my_dsyev_(const char *jobtype, double *A, double *eigval, const int *N, int len)
{
    __linearalgebra_MOD_my_dsyev(jobtype, A, eigval, N, len);
}

} // extern "C"
