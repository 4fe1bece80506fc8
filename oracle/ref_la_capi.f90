!ref_la_capi.f90 -- C-callable doors into the REFERENCE's LinearAlgebra module (test infrastructure only).
!
!This file is NOT a copy of reference code: it only `use`s the module that oracle/build_ref.sh compiles,
!unmodified, from /root/reference/source/LinearAlgebra.f90, and forwards each bind(C) entry to the reference
!routine of the same name (file:line of the routine behind every door is given).  The resulting library
!oracle/_ref/libfl_ref_la.so is used by tools/make_la_golden.py to write tests/golden/la_ref.npz and, where
!/root/reference exists, by tests/test_oracle_pins.py to check the C restatement against the reference itself.
!Nothing of the product links or loads it.
module ref_la_capi
    use iso_c_binding
    use LinearAlgebra
    implicit none
contains
    !My_dpotri  LinearAlgebra.f90:798-812 (dpotrf + dpotri 'L'); the strict upper triangle stays as it was
    subroutine ref_my_dpotri(A,N,info) bind(C,name='ref_my_dpotri')
        integer(c_int),value::N
        real(c_double),intent(inout)::A(N,N)
        integer(c_int),intent(out)::info
        integer::i
        call My_dpotri(A,N,i); info=i
    end subroutine
    !My_dposv  LinearAlgebra.f90:719-730
    subroutine ref_my_dposv(A,b,N,info) bind(C,name='ref_my_dposv')
        integer(c_int),value::N
        real(c_double),intent(inout)::A(N,N),b(N)
        integer(c_int),intent(out)::info
        integer::i
        call My_dposv(A,b,N,i); info=i
    end subroutine
    !My_dsysv  LinearAlgebra.f90:695-703 (no info argument in the reference)
    subroutine ref_my_dsysv(A,b,N) bind(C,name='ref_my_dsysv')
        integer(c_int),value::N
        real(c_double),intent(inout)::A(N,N),b(N)
        call My_dsysv(A,b,N)
    end subroutine
    !vector_direct_product  LinearAlgebra.f90:105-114
    subroutine ref_vector_direct_product(a,b,C,M,N) bind(C,name='ref_vector_direct_product')
        integer(c_int),value::M,N
        real(c_double),intent(in)::a(M),b(N)
        real(c_double),intent(out)::C(M,N)
        C=vector_direct_product(a,b,M,N)
    end subroutine
    !sycp  LinearAlgebra.f90:241-249 (A = lower triangle of B; the rest of A is left alone)
    subroutine ref_sycp(A,B,N) bind(C,name='ref_sycp')
        integer(c_int),value::N
        real(c_double),intent(inout)::A(N,N)
        real(c_double),intent(in)::B(N,N)
        call sycp(A,B,N)
    end subroutine
    !dsyL2U  LinearAlgebra.f90:260-265
    subroutine ref_dsyl2u(A,N) bind(C,name='ref_dsyl2u')
        integer(c_int),value::N
        real(c_double),intent(inout)::A(N,N)
        call dsyL2U(A,N)
    end subroutine
    !My_dgemm  LinearAlgebra.f90:182-188 : C(M,N) = A(M,K) B(K,N)
    subroutine ref_my_dgemm(A,B,C,M,K,N) bind(C,name='ref_my_dgemm')
        integer(c_int),value::M,K,N
        real(c_double),intent(in)::A(M,K),B(K,N)
        real(c_double),intent(out)::C(M,N)
        call My_dgemm(A,B,C,M,K,N)
    end subroutine
    !My_dgemm_T  LinearAlgebra.f90:190-196 : C(M,N) = A(K,M)^T B(K,N)
    subroutine ref_my_dgemm_t(A,B,C,M,K,N) bind(C,name='ref_my_dgemm_t')
        integer(c_int),value::M,K,N
        real(c_double),intent(in)::A(K,M),B(K,N)
        real(c_double),intent(out)::C(M,N)
        call My_dgemm_T(A,B,C,M,K,N)
    end subroutine
    !My_dsyev  LinearAlgebra.f90:879-887 : jobtype 'N' | 'V', lower triangle, ascending eigenvalues
    subroutine ref_my_dsyev(jobtype,A,eigval,N) bind(C,name='ref_my_dsyev')
        character(kind=c_char),value::jobtype
        integer(c_int),value::N
        real(c_double),intent(inout)::A(N,N)
        real(c_double),intent(out)::eigval(N)
        call My_dsyev(jobtype,A,eigval,N)
    end subroutine
end module ref_la_capi
