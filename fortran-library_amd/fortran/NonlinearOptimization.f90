!Nonlinear optimization routines -- MI355X build of the line-search optimisers
!
!Drop-in replacement for the line-search part of the reference module
!(/root/reference/source/NonlinearOptimization.f90): same module name, same procedure
!names, same dummy-argument names (keyword calls keep working), same optional-ness,
!same callback conventions:
!    subroutine f(f(x), x, dim)
!    subroutine fd(f'(x), x, dim)
!    integer function f_fd(f(x), f'(x), x, dim)
!This file contains interfaces and argument marshalling only: every procedure forwards
!through iso_c_binding to libFL.so (include/fl_legacy.h, include/fl_nlopt.h), where the
!solver runs on the GPU and calls f / fd / f_fd back on the host when it needs them.
!  SteepestDescent   <- reference NonlinearOptimization.f90:55
!  ConjugateGradient <- reference NonlinearOptimization.f90:193 (Method = 'DY' | 'PR')
!  LBFGS             <- reference NonlinearOptimization.f90:398
!  BFGS              <- reference NonlinearOptimization.f90:632 (ExactStep > 0: Hessian from fdd or central differences)
!New: LBFGS_batched / ConjugateGradient_batched -- batches of independent problems with
!device-resident data and built-in objectives (include/fl_nlopt.h).
!  NewtonRaphson     <- reference NonlinearOptimization.f90:1026 (without fdd: central differences of fd)
!  AugmentedLagrangian <- reference NonlinearOptimization.f90:2005 (inner solvers LBFGS / ConjugateGradient / BFGS / NewtonRaphson)
!  Wolfe, Wolfe_fdwithf, StrongWolfe, StrongWolfe_fdwithf <- reference NonlinearOptimization.f90:1286, 1373, 1462, 1582
!  LagrangianMultiplier <- reference NonlinearOptimization.f90:1950
!  TrustRegion, TrustRegion_basic <- reference NonlinearOptimization.f90:1728, 2348 (own Levenberg-Marquardt
!      iteration behind the interface of the MKL trnlsp wrapper)
module NonlinearOptimization
    use iso_c_binding
    implicit none

    integer(c_int),parameter::FL_OBJ_QUARTIC=0,FL_OBJ_ROSENBROCK=1,FL_OBJ_DIAGQUAD=2
    integer(c_int),parameter::FL_SOLVER_SD=0,FL_SOLVER_CG=1,FL_SOLVER_LBFGS=2,FL_SOLVER_BFGS=3,FL_SOLVER_NEWTON=4
    integer(c_int),parameter::FL_REQ_F=1,FL_REQ_G=2,FL_REQ_SAME=4,FL_RCI_BOTH=1,FL_RCI_REQ_C=32,FL_RCI_REQ_CD=64

    type,bind(C)::fl_options!include/fl_nlopt.h: struct fl_options
        integer(c_int32_t)::strong,max_iteration
        real(c_double)::precision,min_step_length,wolfe_c1,wolfe_c2,increment
        integer(c_int32_t)::memory,cg_method,fused_f_fd,clamp,exact_step
    end type fl_options

    interface
        subroutine flc_steepestdescent(f,fd,x,dim,f_fd,Strong,Warning,MaxIteration,Precision,MinStepLength,&
        WolfeConst1,WolfeConst2,Increment) bind(C,name='__nonlinearoptimization_MOD_steepestdescent')
            import
            type(c_funptr),value::f,fd,f_fd
            real(c_double)::x(*)
            integer(c_int),intent(in)::dim
            type(c_ptr),value::Strong,Warning,MaxIteration,Precision,MinStepLength,WolfeConst1,WolfeConst2,Increment
        end subroutine flc_steepestdescent
        subroutine flc_conjugategradient(f,fd,x,dim,Method,f_fd,Strong,Warning,MaxIteration,Precision,MinStepLength,&
        WolfeConst1,WolfeConst2,Increment,len_Method) bind(C,name='__nonlinearoptimization_MOD_conjugategradient')
            import
            type(c_funptr),value::f,fd,f_fd
            real(c_double)::x(*)
            integer(c_int),intent(in)::dim
            character(kind=c_char)::Method(*)
            type(c_ptr),value::Strong,Warning,MaxIteration,Precision,MinStepLength,WolfeConst1,WolfeConst2,Increment
            integer(c_int),value::len_Method
        end subroutine flc_conjugategradient
        subroutine flc_lbfgs(f,fd,x,dim,Memory,f_fd,Strong,Warning,MaxIteration,Precision,MinStepLength,&
        WolfeConst1,WolfeConst2,Increment) bind(C,name='__nonlinearoptimization_MOD_lbfgs')
            import
            type(c_funptr),value::f,fd,f_fd
            real(c_double)::x(*)
            integer(c_int),intent(in)::dim
            type(c_ptr),value::Memory,Strong,Warning,MaxIteration,Precision,MinStepLength,WolfeConst1,WolfeConst2,Increment
        end subroutine flc_lbfgs
        subroutine flc_bfgs(f,fd,x,dim,fdd,ExactStep,f_fd,Strong,Warning,MaxIteration,Precision,MinStepLength,&
        WolfeConst1,WolfeConst2,Increment) bind(C,name='__nonlinearoptimization_MOD_bfgs')
            import
            type(c_funptr),value::f,fd,fdd,f_fd
            real(c_double)::x(*)
            integer(c_int),intent(in)::dim
            type(c_ptr),value::ExactStep,Strong,Warning,MaxIteration,Precision,MinStepLength,WolfeConst1,WolfeConst2,Increment
        end subroutine flc_bfgs
        subroutine flc_newtonraphson(f,fd,x,dim,fdd,f_fd,Strong,Warning,MaxIteration,Precision,MinStepLength,&
        WolfeConst1,WolfeConst2,Increment) bind(C,name='__nonlinearoptimization_MOD_newtonraphson')
            import
            type(c_funptr),value::f,fd,fdd,f_fd
            real(c_double)::x(*)
            integer(c_int),intent(in)::dim
            type(c_ptr),value::Strong,Warning,MaxIteration,Precision,MinStepLength,WolfeConst1,WolfeConst2,Increment
        end subroutine flc_newtonraphson
        subroutine flc_augmentedlagrangian(f,fd,c,cd,x,N,M,UnconstrainedSolver,lambda0,miu0,fdd,cdd,ExactStep,Memory,&
        Method,f_fd,Strong,Warning,MaxIteration,Precision,MinStepLength,WolfeConst1,WolfeConst2,Increment,&
        len_UnconstrainedSolver,len_Method) bind(C,name='__nonlinearoptimization_MOD_augmentedlagrangian')
            import
            type(c_funptr),value::f,fd,c,cd,fdd,cdd,f_fd
            real(c_double)::x(*)
            integer(c_int),intent(in)::N,M
            character(kind=c_char)::UnconstrainedSolver(*),Method(*)
            type(c_ptr),value::lambda0,miu0,ExactStep,Memory
            type(c_ptr),value::Strong,Warning,MaxIteration,Precision,MinStepLength,WolfeConst1,WolfeConst2,Increment
            integer(c_int),value::len_UnconstrainedSolver,len_Method
        end subroutine flc_augmentedlagrangian
        subroutine flc_linesearch(c1,c2,f,fd,x,a,p,fx,phid0,fdx,dim,Increment) bind(C,name='__nonlinearoptimization_MOD_wolfe')
            import
            real(c_double),intent(in)::c1,c2,phid0
            type(c_funptr),value::f,fd
            real(c_double)::x(*),fdx(*)
            real(c_double),intent(in)::p(*)
            real(c_double),intent(inout)::a,fx
            integer(c_int),intent(in)::dim
            type(c_ptr),value::Increment
        end subroutine flc_linesearch
        subroutine flc_stronglinesearch(c1,c2,f,fd,f_fd,x,a,p,fx,phid0,fdx,dim,Increment)&
        bind(C,name='__nonlinearoptimization_MOD_strongwolfe_fdwithf')
            import
            real(c_double),intent(in)::c1,c2,phid0
            type(c_funptr),value::f,fd,f_fd
            real(c_double)::x(*),fdx(*)
            real(c_double),intent(in)::p(*)
            real(c_double),intent(inout)::a,fx
            integer(c_int),intent(in)::dim
            type(c_ptr),value::Increment
        end subroutine flc_stronglinesearch
        subroutine flc_lagrangianmultiplier(fd,fdd,c,cd,cdd,x,lambda,N,M,Warning,MaxIteration,Precision)&
        bind(C,name='__nonlinearoptimization_MOD_lagrangianmultiplier')
            import
            type(c_funptr),value::fd,fdd,c,cd,cdd
            real(c_double)::x(*),lambda(*)
            integer(c_int),intent(in)::N,M
            type(c_ptr),value::Warning,MaxIteration,Precision
        end subroutine flc_lagrangianmultiplier
        subroutine flc_trustregion(fd,x,M,N,Jacobian,low,up,Warning,MaxIteration,MaxStepIteration,Precision,MinStepLength)&
        bind(C,name='__nonlinearoptimization_MOD_trustregion')
            import
            type(c_funptr),value::fd,Jacobian
            real(c_double)::x(*)
            integer(c_int),intent(in)::M,N
            type(c_ptr),value::low,up,Warning,MaxIteration,MaxStepIteration,Precision,MinStepLength
        end subroutine flc_trustregion
        subroutine fl_default_options(opt,solver) bind(C,name='fl_default_options')
            import
            type(fl_options),intent(out)::opt
            integer(c_int),value::solver
        end subroutine fl_default_options
        integer(c_size_t) function fl_workspace_bytes(solver,batch,n,memory) bind(C,name='fl_workspace_bytes')
            import
            integer(c_int),value::solver,batch,n,memory
        end function fl_workspace_bytes
        integer(c_int) function fl_lbfgs_batched(objective,batch,n,x,d,b,opt,ws,ws_bytes,f,gg,iters,status,nf,ng,stream)&
        bind(C,name='fl_lbfgs_batched')
            import
            integer(c_int),value::objective,batch,n
            type(c_ptr),value::x,d,b,ws,f,gg,iters,status,nf,ng,stream!device pointers
            type(fl_options),intent(in)::opt
            integer(c_size_t),value::ws_bytes
        end function fl_lbfgs_batched
        integer(c_int) function fl_conjugate_gradient_batched(objective,batch,n,x,d,b,opt,f,gg,iters,status,nf,ng,stream)&
        bind(C,name='fl_conjugate_gradient_batched')
            import
            integer(c_int),value::objective,batch,n
            type(c_ptr),value::x,d,b,f,gg,iters,status,nf,ng,stream
            type(fl_options),intent(in)::opt
        end function fl_conjugate_gradient_batched
        !A batch of independent problems over all the GPUs of the node: host arrays x(n,batch), d(n,batch), b(n,batch) in,
        !results out, one host thread per shard (include/fl_nlopt.h: fl_multi_solve).  Absent outputs: c_null_ptr.
        integer(c_int) function fl_multi_device_count() bind(C,name='fl_multi_device_count')
            import
        end function fl_multi_device_count
        integer(c_int) function fl_multi_solve(solver,objective,batch,n,x,d,b,opt,aug_m,lambda,miu0,f,gg,cnorm2,iters,outer,&
        status,nf,ng,nshards,interleaved) bind(C,name='fl_multi_solve')
            import
            integer(c_int),value::solver,objective,batch,n,aug_m,nshards,interleaved
            type(c_ptr),value::x,d,b,lambda,f,gg,cnorm2,iters,outer,status,nf,ng!HOST pointers (c_loc of the arrays)
            type(fl_options),intent(in)::opt
            real(c_double),value::miu0
        end function fl_multi_solve
        !---- the other batched solvers on device-resident data (include/fl_nlopt.h; all pointers: device, c_null_ptr = absent)
        integer(c_size_t) function fl_workspace_bytes_for(solver,batch,n,opt) bind(C,name='fl_workspace_bytes_for')
            import
            integer(c_int),value::solver,batch,n
            type(fl_options),intent(in)::opt
        end function fl_workspace_bytes_for
        integer(c_int) function fl_steepest_descent_batched(objective,batch,n,x,d,b,opt,f,gg,iters,status,nf,ng,stream)&
        bind(C,name='fl_steepest_descent_batched')
            import
            integer(c_int),value::objective,batch,n
            type(c_ptr),value::x,d,b,f,gg,iters,status,nf,ng,stream
            type(fl_options),intent(in)::opt
        end function fl_steepest_descent_batched
        integer(c_int) function fl_bfgs_batched(objective,batch,n,x,d,b,opt,ws,ws_bytes,f,gg,iters,status,nf,ng,stream)&
        bind(C,name='fl_bfgs_batched')
            import
            integer(c_int),value::objective,batch,n
            type(c_ptr),value::x,d,b,ws,f,gg,iters,status,nf,ng,stream
            type(fl_options),intent(in)::opt
            integer(c_size_t),value::ws_bytes
        end function fl_bfgs_batched
        integer(c_int) function fl_newton_raphson_batched(objective,batch,n,x,d,b,opt,ws,ws_bytes,f,gg,iters,status,nf,ng,stream)&
        bind(C,name='fl_newton_raphson_batched')
            import
            integer(c_int),value::objective,batch,n
            type(c_ptr),value::x,d,b,ws,f,gg,iters,status,nf,ng,stream
            type(fl_options),intent(in)::opt
            integer(c_size_t),value::ws_bytes
        end function fl_newton_raphson_batched
        integer(c_int) function fl_augmented_lagrangian_batched(solver,objective,batch,n,m,x,d,b,lambda,miu0,opt,ws,ws_bytes,f,&
        cnorm2,iters,outer,status,nf,ng,stream) bind(C,name='fl_augmented_lagrangian_batched')
            import
            integer(c_int),value::solver,objective,batch,n,m
            type(c_ptr),value::x,d,b,lambda,ws,f,cnorm2,iters,outer,status,nf,ng,stream
            real(c_double),value::miu0
            type(fl_options),intent(in)::opt
            integer(c_size_t),value::ws_bytes
        end function fl_augmented_lagrangian_batched
        !---- reverse communication: the caller evaluates f, f' (and c, cd) for the batch between two steps -- the reference's
        !callbacks f, fd, c, cd (NonlinearOptimization.f90:33-38, 1928-1934) as an ask / tell loop (include/fl_nlopt.h)
        integer(c_int) function fl_rci_create(handle,solver,batch,n,opt,stream) bind(C,name='fl_rci_create')
            import
            type(c_ptr),intent(out)::handle
            integer(c_int),value::solver,batch,n
            type(fl_options),intent(in)::opt
            type(c_ptr),value::stream
        end function fl_rci_create
        integer(c_int) function fl_rci_step(handle,x,f,g,request) bind(C,name='fl_rci_step')
            import
            type(c_ptr),value::handle,x,f,g,request
        end function fl_rci_step
        integer(c_int) function fl_rci_step_flags(handle,x,f,g,request,flags) bind(C,name='fl_rci_step_flags')
            import
            type(c_ptr),value::handle,x,f,g,request
            integer(c_int),value::flags
        end function fl_rci_step_flags
        integer(c_int) function fl_rci_results(handle,f,gg,iters,status,nf,ng) bind(C,name='fl_rci_results')
            import
            type(c_ptr),value::handle,f,gg,iters,status,nf,ng
        end function fl_rci_results
        integer(c_int) function fl_rci_destroy(handle) bind(C,name='fl_rci_destroy')
            import
            type(c_ptr),value::handle
        end function fl_rci_destroy
        integer(c_int) function fl_rci_create_auglag(handle,solver,batch,n,m,lambda,miu0,opt,stream) bind(C,name='fl_rci_create_auglag')
            import
            type(c_ptr),intent(out)::handle
            integer(c_int),value::solver,batch,n,m
            type(c_ptr),value::lambda,stream
            real(c_double),value::miu0
            type(fl_options),intent(in)::opt
        end function fl_rci_create_auglag
        integer(c_int) function fl_rci_step_auglag(handle,x,f,g,c,cd,request) bind(C,name='fl_rci_step_auglag')
            import
            type(c_ptr),value::handle,x,f,g,c,cd,request
        end function fl_rci_step_auglag
        integer(c_int) function fl_rci_results_auglag(handle,cnorm2,outer) bind(C,name='fl_rci_results_auglag')
            import
            type(c_ptr),value::handle,cnorm2,outer
        end function fl_rci_results_auglag
        !---- the caller's objective as HIP source text, compiled into the fused kernel at run time (include/fl_nlopt.h):
        !source and class_name are C strings (trim(text)//c_null_char); log receives the compiler's messages
        integer(c_int) function fl_user_compile(handle,source,class_name,solver,n,tune_like,log,log_bytes) bind(C,name='fl_user_compile')
            import
            type(c_ptr),intent(out)::handle
            character(kind=c_char),intent(in)::source(*),class_name(*)
            integer(c_int),value::solver,n,tune_like
            character(kind=c_char)::log(*)
            integer(c_size_t),value::log_bytes
        end function fl_user_compile
        integer(c_int) function fl_user_solve(handle,batch,x,data0,data1,params,opt,ws,ws_bytes,f,gg,iters,status,nf,ng,stream)&
        bind(C,name='fl_user_solve')
            import
            type(c_ptr),value::handle,x,data0,data1,params,ws,f,gg,iters,status,nf,ng,stream
            integer(c_int),value::batch
            type(fl_options),intent(in)::opt
            integer(c_size_t),value::ws_bytes
        end function fl_user_solve
        integer(c_int) function fl_user_destroy(handle) bind(C,name='fl_user_destroy')
            import
            type(c_ptr),value::handle
        end function fl_user_destroy
        !workgroups that share one problem of this batch in the fused solvers (n > 14336, few problems; 1: none)
        integer(c_int) function fl_cooperative_groups_for(solver,objective,batch,n) bind(C,name='fl_cooperative_groups_for')
            import
            integer(c_int),value::solver,objective,batch,n
        end function fl_cooperative_groups_for
    end interface

contains
!-------------- Line search --------------
    !Optional arguments travel as C pointers: the address of a local copy when present, NULL when absent
    !(the convention of the reference's C++ header, cpp/README.md:11-18); logical -> 4-byte integer

    !Line searchers (reference NonlinearOptimization.f90:1286, 1373, 1462, 1582): public in the reference module.
    !Input:  Wolfe constants c1 & c2, x, initial guess a, direction p, fx = f(x), phid0 = phi'(0)
    !Output: a = accepted step, x = x + a * p, fx = f(x), fdx = f'(x)
    subroutine Wolfe(c1, c2, f, fd, x, a, p, fx, phid0, fdx, dim, Increment)
        real*8,intent(in)::c1,c2
        external::f,fd
        integer,intent(in)::dim
        real*8,dimension(dim),intent(inout)::x
        real*8,intent(inout)::a
        real*8,dimension(dim),intent(in)::p
        real*8,intent(inout)::fx
        real*8,intent(in)::phid0
        real*8,dimension(dim),intent(out)::fdx
        real*8,intent(in),optional::Increment
        real(c_double),target::li
        type(c_ptr)::pinc
        pinc=c_null_ptr; if(present(Increment)) then; li=Increment; pinc=c_loc(li); end if
        call flc_linesearch(c1,c2,c_funloc(f),c_funloc(fd),x,a,p,fx,phid0,fdx,dim,pinc)
    end subroutine Wolfe
    subroutine Wolfe_fdwithf(c1, c2, f, fd, f_fd, x, a, p, fx, phid0, fdx, dim, Increment)!never calls f_fd (reference 1373)
        real*8,intent(in)::c1,c2
        external::f,fd
        integer,external::f_fd
        integer,intent(in)::dim
        real*8,dimension(dim),intent(inout)::x
        real*8,intent(inout)::a
        real*8,dimension(dim),intent(in)::p
        real*8,intent(inout)::fx
        real*8,intent(in)::phid0
        real*8,dimension(dim),intent(out)::fdx
        real*8,intent(in),optional::Increment
        call Wolfe(c1,c2,f,fd,x,a,p,fx,phid0,fdx,dim,Increment)
    end subroutine Wolfe_fdwithf
    subroutine StrongWolfe(c1, c2, f, fd, x, a, p, fx, phid0, fdx, dim, Increment)
        real*8,intent(in)::c1,c2
        external::f,fd
        integer,intent(in)::dim
        real*8,dimension(dim),intent(inout)::x
        real*8,intent(inout)::a
        real*8,dimension(dim),intent(in)::p
        real*8,intent(inout)::fx
        real*8,intent(in)::phid0
        real*8,dimension(dim),intent(out)::fdx
        real*8,intent(in),optional::Increment
        real(c_double),target::li
        type(c_ptr)::pinc
        pinc=c_null_ptr; if(present(Increment)) then; li=Increment; pinc=c_loc(li); end if
        call flc_stronglinesearch(c1,c2,c_funloc(f),c_funloc(fd),c_null_funptr,x,a,p,fx,phid0,fdx,dim,pinc)
    end subroutine StrongWolfe
    subroutine StrongWolfe_fdwithf(c1, c2, f, fd, f_fd, x, a, p, fx, phid0, fdx, dim, Increment)
        real*8,intent(in)::c1,c2
        external::f,fd
        integer,external::f_fd
        integer,intent(in)::dim
        real*8,dimension(dim),intent(inout)::x
        real*8,intent(inout)::a
        real*8,dimension(dim),intent(in)::p
        real*8,intent(inout)::fx
        real*8,intent(in)::phid0
        real*8,dimension(dim),intent(out)::fdx
        real*8,intent(in),optional::Increment
        real(c_double),target::li
        type(c_ptr)::pinc
        pinc=c_null_ptr; if(present(Increment)) then; li=Increment; pinc=c_loc(li); end if
        call flc_stronglinesearch(c1,c2,c_funloc(f),c_funloc(fd),c_funloc(f_fd),x,a,p,fx,phid0,fdx,dim,pinc)
    end subroutine StrongWolfe_fdwithf

    subroutine SteepestDescent(f, fd, x, dim, &
    f_fd, Strong, Warning, MaxIteration, Precision, MinStepLength, WolfeConst1, WolfeConst2, Increment)
        external::f,fd
        integer,intent(in)::dim
        real*8,dimension(dim),intent(inout)::x
        integer,external,optional::f_fd
        logical,intent(in),optional::Strong,Warning
        integer,intent(in),optional::MaxIteration
        real*8,intent(in),optional::Precision,MinStepLength,WolfeConst1,WolfeConst2,Increment
        integer(c_int32_t),target::ls(2)
        integer(c_int),target::li(1)
        real(c_double),target::lr(5)
        type(c_ptr)::p(8)
        type(c_funptr)::pf_fd
        call pack_common(p,ls,li,lr,Strong,Warning,MaxIteration,Precision,MinStepLength,WolfeConst1,WolfeConst2,Increment)
        pf_fd=c_null_funptr; if(present(f_fd)) pf_fd=c_funloc(f_fd)
        call flc_steepestdescent(c_funloc(f),c_funloc(fd),x,dim,pf_fd,p(1),p(2),p(3),p(4),p(5),p(6),p(7),p(8))
    end subroutine SteepestDescent

    subroutine ConjugateGradient(f, fd, x, dim, &
    Method, &
    f_fd, Strong, Warning, MaxIteration, Precision, MinStepLength, WolfeConst1, WolfeConst2, Increment)
        external::f,fd
        integer,intent(in)::dim
        real*8,dimension(dim),intent(inout)::x
        character(*),intent(in),optional::Method
        integer,external,optional::f_fd
        logical,intent(in),optional::Strong,Warning
        integer,intent(in),optional::MaxIteration
        real*8,intent(in),optional::Precision,MinStepLength,WolfeConst1,WolfeConst2,Increment
        integer(c_int32_t),target::ls(2)
        integer(c_int),target::li(1)
        real(c_double),target::lr(5)
        type(c_ptr)::p(8)
        type(c_funptr)::pf_fd
        character(kind=c_char)::m(2)
        m(1)='D'; m(2)='Y'!default = DY (reference NonlinearOptimization.f90:214-215)
        if(present(Method)) then
            m=' '
            if(len(Method)>=1) m(1)=Method(1:1)
            if(len(Method)>=2) m(2)=Method(2:2)
        end if
        call pack_common(p,ls,li,lr,Strong,Warning,MaxIteration,Precision,MinStepLength,WolfeConst1,WolfeConst2,Increment)
        pf_fd=c_null_funptr; if(present(f_fd)) pf_fd=c_funloc(f_fd)
        call flc_conjugategradient(c_funloc(f),c_funloc(fd),x,dim,m,pf_fd,p(1),p(2),p(3),p(4),p(5),p(6),p(7),p(8),2_c_int)
    end subroutine ConjugateGradient

    subroutine LBFGS(f, fd, x, dim, &
    Memory, &
    f_fd, Strong, Warning, MaxIteration, Precision, MinStepLength, WolfeConst1, WolfeConst2, Increment)
        external::f,fd
        integer,intent(in)::dim
        real*8,dimension(dim),intent(inout)::x
        integer,external,optional::f_fd
        logical,intent(in),optional::Strong,Warning
        integer,intent(in),optional::Memory,MaxIteration
        real*8,intent(in),optional::Precision,MinStepLength,WolfeConst1,WolfeConst2,Increment
        integer(c_int32_t),target::ls(2)
        integer(c_int),target::li(1),lm
        real(c_double),target::lr(5)
        type(c_ptr)::p(8),pm
        type(c_funptr)::pf_fd
        call pack_common(p,ls,li,lr,Strong,Warning,MaxIteration,Precision,MinStepLength,WolfeConst1,WolfeConst2,Increment)
        pm=c_null_ptr
        if(present(Memory)) then; lm=Memory; pm=c_loc(lm); end if
        pf_fd=c_null_funptr; if(present(f_fd)) pf_fd=c_funloc(f_fd)
        call flc_lbfgs(c_funloc(f),c_funloc(fd),x,dim,pm,pf_fd,p(1),p(2),p(3),p(4),p(5),p(6),p(7),p(8))
    end subroutine LBFGS

    subroutine BFGS(f, fd, x, dim, &
    fdd, ExactStep, &
    f_fd, Strong, Warning, MaxIteration, Precision, MinStepLength, WolfeConst1, WolfeConst2, Increment)
        external::f,fd
        integer,intent(in)::dim
        real*8,dimension(dim),intent(inout)::x
        integer,external,optional::fdd,f_fd
        logical,intent(in),optional::Strong,Warning
        integer,intent(in),optional::ExactStep,MaxIteration
        real*8,intent(in),optional::Precision,MinStepLength,WolfeConst1,WolfeConst2,Increment
        integer(c_int32_t),target::ls(2)
        integer(c_int),target::li(1),le
        real(c_double),target::lr(5)
        type(c_ptr)::p(8),pe
        type(c_funptr)::pf_fd,pfdd
        call pack_common(p,ls,li,lr,Strong,Warning,MaxIteration,Precision,MinStepLength,WolfeConst1,WolfeConst2,Increment)
        pe=c_null_ptr
        if(present(ExactStep)) then; le=ExactStep; pe=c_loc(le); end if
        pf_fd=c_null_funptr; if(present(f_fd)) pf_fd=c_funloc(f_fd)
        pfdd=c_null_funptr; if(present(fdd)) pfdd=c_funloc(fdd)
        call flc_bfgs(c_funloc(f),c_funloc(fd),x,dim,pfdd,pe,pf_fd,p(1),p(2),p(3),p(4),p(5),p(6),p(7),p(8))
    end subroutine BFGS

    !Newton-Raphson method (reference NonlinearOptimization.f90:1026); without fdd the Hessian comes from central
    !differences of fd (the reference calls MKL djacobi there)
    subroutine NewtonRaphson(f, fd, x, dim, &
    fdd, &
    f_fd, Strong, Warning, MaxIteration, Precision, MinStepLength, WolfeConst1, WolfeConst2, Increment)
        external::f,fd
        integer,intent(in)::dim
        real*8,dimension(dim),intent(inout)::x
        integer,external,optional::fdd,f_fd
        logical,intent(in),optional::Strong,Warning
        integer,intent(in),optional::MaxIteration
        real*8,intent(in),optional::Precision,MinStepLength,WolfeConst1,WolfeConst2,Increment
        integer(c_int32_t),target::ls(2)
        integer(c_int),target::li(1)
        real(c_double),target::lr(5)
        type(c_ptr)::p(8)
        type(c_funptr)::pf_fd,pfdd
        call pack_common(p,ls,li,lr,Strong,Warning,MaxIteration,Precision,MinStepLength,WolfeConst1,WolfeConst2,Increment)
        pf_fd=c_null_funptr; if(present(f_fd)) pf_fd=c_funloc(f_fd)
        pfdd=c_null_funptr; if(present(fdd)) pfdd=c_funloc(fdd)
        call flc_newtonraphson(c_funloc(f),c_funloc(fd),x,dim,pfdd,pf_fd,p(1),p(2),p(3),p(4),p(5),p(6),p(7),p(8))
    end subroutine NewtonRaphson

    !Trust region (reference NonlinearOptimization.f90:1728-1906, 2348-2423): solve f'(x) = 0 by minimising |f'(x)|^2.
    !The reference wraps MKL's closed trnlsp solver; behind the same interface this build runs an own
    !Levenberg-Marquardt iteration (callbacks on the host, normal equations on the device): same stationary points
    !    subroutine fd(f'(x), x, M, N);  integer function Jacobian(J(x), x, M, N), M x N; M >= N
    subroutine TrustRegion(fd, x, M, N, &
    Jacobian, low, up, Warning, MaxIteration, MaxStepIteration, Precision, MinStepLength)
        external::fd
        integer,intent(in)::M,N
        real*8,dimension(N),intent(inout)::x
        integer,external,optional::Jacobian
        real*8,dimension(N),intent(in),optional,target::low,up
        logical,intent(in),optional::Warning
        integer,intent(in),optional::MaxIteration,MaxStepIteration
        real*8 ,intent(in),optional::Precision,MinStepLength
        integer(c_int32_t),target::lw
        integer(c_int),target::lmax,lstep
        real(c_double),target::lp,lmin
        type(c_ptr)::pl,pu,pw,pm,ps,pp,pn
        type(c_funptr)::pj
        pj=c_null_funptr; if(present(Jacobian)) pj=c_funloc(Jacobian)
        pl=c_null_ptr; pu=c_null_ptr
        if(present(low).and.present(up)) then; pl=c_loc(low); pu=c_loc(up); end if
        pw=c_null_ptr; if(present(Warning)) then; lw=merge(1,0,Warning); pw=c_loc(lw); end if
        pm=c_null_ptr; if(present(MaxIteration)) then; lmax=MaxIteration; pm=c_loc(lmax); end if
        ps=c_null_ptr; if(present(MaxStepIteration)) then; lstep=MaxStepIteration; ps=c_loc(lstep); end if
        pp=c_null_ptr; if(present(Precision)) then; lp=Precision; pp=c_loc(lp); end if
        pn=c_null_ptr; if(present(MinStepLength)) then; lmin=MinStepLength; pn=c_loc(lmin); end if
        call flc_trustregion(c_funloc(fd),x,M,N,pj,pl,pu,pw,pm,ps,pp,pn)
    end subroutine TrustRegion
    subroutine TrustRegion_basic(fd, Jacobian, x, M, N, &
    Warning, MaxIteration, MaxStepIteration, Precision, MinStepLength)
        external::fd
        integer,external::Jacobian
        integer,intent(in)::M,N
        real*8,dimension(N),intent(inout)::x
        logical,intent(in)::Warning
        integer,intent(in)::MaxIteration,MaxStepIteration
        real*8 ,intent(in)::Precision,MinStepLength
        call TrustRegion(fd,x,M,N,Jacobian=Jacobian,Warning=Warning,MaxIteration=MaxIteration,&
            MaxStepIteration=MaxStepIteration,Precision=Precision,MinStepLength=MinStepLength)
    end subroutine TrustRegion_basic

    !Lagrangian multiplier method (reference NonlinearOptimization.f90:1950-1993): Newton iteration on the KKT system;
    !on input lambda is an initial guess of the multipliers, on exit the solution.  fd, fdd, c, cd, cdd are evaluated on
    !the host, the (N+M)-dimensional symmetric indefinite solve (My_dsysv) runs on the device
    subroutine LagrangianMultiplier(fd, fdd, c, cd, cdd, x, lambda, N, M, &
    Warning, MaxIteration, Precision)
        external::fd,c,cd; integer,external::fdd,cdd
        integer,intent(in)::N,M
        real*8,dimension(N),intent(inout)::x; real*8,dimension(M),intent(inout)::lambda
        logical,intent(in),optional::Warning
        integer,intent(in),optional::MaxIteration
        real*8,intent(in),optional::Precision
        integer(c_int32_t),target::lw
        integer(c_int),target::lmax
        real(c_double),target::lp
        type(c_ptr)::pw,pm,pp
        pw=c_null_ptr; if(present(Warning)) then; lw=merge(1,0,Warning); pw=c_loc(lw); end if
        pm=c_null_ptr; if(present(MaxIteration)) then; lmax=MaxIteration; pm=c_loc(lmax); end if
        pp=c_null_ptr; if(present(Precision)) then; lp=Precision; pp=c_loc(lp); end if
        call flc_lagrangianmultiplier(c_funloc(fd),c_funloc(fdd),c_funloc(c),c_funloc(cd),c_funloc(cdd),x,lambda,N,M,pw,pm,pp)
    end subroutine LagrangianMultiplier

    !Augmented Lagrangian multiplier method (reference NonlinearOptimization.f90:2005-2241): equality constraints
    !c(x)=0 with  subroutine c(c(x),x,M,N),  subroutine cd(c'(x),x,M,N) (c'(x) is N x M).  Inner solvers on the device:
    !'LBFGS', 'ConjugateGradient', 'BFGS' (default), 'NewtonRaphson'; the wrappers L, Ld, Ldd (2193-2240) are evaluated on
    !the host next to the caller's f, fd, c, cd (fdd, cdd); without fdd & cdd the Hessian of L is differentiated numerically
    subroutine AugmentedLagrangian(f, fd, c, cd, x, N, M, &
    UnconstrainedSolver, lambda0, miu0, &
    fdd, cdd, ExactStep, Memory, Method, &
    f_fd, Strong, Warning, MaxIteration, Precision, MinStepLength, WolfeConst1, WolfeConst2, Increment)
        external::f,fd,c,cd
        integer,intent(in)::N,M
        real*8,dimension(N),intent(inout)::x
        character(*),intent(in),optional::UnconstrainedSolver
        real*8,dimension(M),intent(in),optional,target::lambda0
        integer,external,optional::f_fd,fdd,cdd
        logical,intent(in),optional::Strong,Warning
        integer,intent(in),optional::MaxIteration,ExactStep,Memory
        real*8,intent(in),optional::miu0,Precision,MinStepLength,WolfeConst1,WolfeConst2,Increment
        character(*),intent(in),optional::Method
        integer(c_int32_t),target::ls(2)
        integer(c_int),target::li(1),le,lm
        real(c_double),target::lr(5),lmiu
        type(c_ptr)::p(8),pe,pm,pl,pmiu
        type(c_funptr)::pf_fd,pfdd,pcdd
        character(kind=c_char)::sv(32),mt(2)
        integer::i,lsv
        sv=' '; lsv=4; sv(1)='B'; sv(2)='F'; sv(3)='G'; sv(4)='S'!default solver (reference NonlinearOptimization.f90:2041)
        if(present(UnconstrainedSolver)) then
            sv=' '; lsv=min(len(UnconstrainedSolver),32)
            do i=1,lsv; sv(i)=UnconstrainedSolver(i:i); end do
        end if
        mt(1)='D'; mt(2)='Y'
        if(present(Method)) then
            mt=' '
            if(len(Method)>=1) mt(1)=Method(1:1)
            if(len(Method)>=2) mt(2)=Method(2:2)
        end if
        call pack_common(p,ls,li,lr,Strong,Warning,MaxIteration,Precision,MinStepLength,WolfeConst1,WolfeConst2,Increment)
        pe=c_null_ptr; if(present(ExactStep)) then; le=ExactStep; pe=c_loc(le); end if
        pm=c_null_ptr; if(present(Memory)) then; lm=Memory; pm=c_loc(lm); end if
        pmiu=c_null_ptr; if(present(miu0)) then; lmiu=miu0; pmiu=c_loc(lmiu); end if
        pl=c_null_ptr; if(present(lambda0)) pl=c_loc(lambda0)
        pf_fd=c_null_funptr; if(present(f_fd)) pf_fd=c_funloc(f_fd)
        pfdd=c_null_funptr; if(present(fdd)) pfdd=c_funloc(fdd)
        pcdd=c_null_funptr; if(present(cdd)) pcdd=c_funloc(cdd)
        call flc_augmentedlagrangian(c_funloc(f),c_funloc(fd),c_funloc(c),c_funloc(cd),x,N,M,sv,pl,pmiu,pfdd,pcdd,pe,pm,&
            mt,pf_fd,p(1),p(2),p(3),p(4),p(5),p(6),p(7),p(8),int(lsv,c_int),2_c_int)
    end subroutine AugmentedLagrangian

    subroutine pack_common(p,ls,li,lr,Strong,Warning,MaxIteration,Precision,MinStepLength,WolfeConst1,WolfeConst2,Increment)
        type(c_ptr),intent(out)::p(8)
        integer(c_int32_t),target,intent(out)::ls(2)
        integer(c_int),target,intent(out)::li(1)
        real(c_double),target,intent(out)::lr(5)
        logical,intent(in),optional::Strong,Warning
        integer,intent(in),optional::MaxIteration
        real*8,intent(in),optional::Precision,MinStepLength,WolfeConst1,WolfeConst2,Increment
        p=c_null_ptr; ls=0; li=0; lr=0d0
        if(present(Strong)) then; ls(1)=merge(1,0,Strong); p(1)=c_loc(ls(1)); end if
        if(present(Warning)) then; ls(2)=merge(1,0,Warning); p(2)=c_loc(ls(2)); end if
        if(present(MaxIteration)) then; li(1)=MaxIteration; p(3)=c_loc(li(1)); end if
        if(present(Precision)) then; lr(1)=Precision; p(4)=c_loc(lr(1)); end if
        if(present(MinStepLength)) then; lr(2)=MinStepLength; p(5)=c_loc(lr(2)); end if
        if(present(WolfeConst1)) then; lr(3)=WolfeConst1; p(6)=c_loc(lr(3)); end if
        if(present(WolfeConst2)) then; lr(4)=WolfeConst2; p(7)=c_loc(lr(4)); end if
        if(present(Increment)) then; lr(5)=Increment; p(8)=c_loc(lr(5)); end if
    end subroutine pack_common

    !Batched L-BFGS on device-resident data: same keyword names and defaults as LBFGS
    subroutine LBFGS_batched(objective, x_dev, batch, dim, d_dev, b_dev, ws_dev, ws_bytes, &
    f_dev, iters_dev, status_dev, info, &
    Memory, Strong, MaxIteration, Precision, MinStepLength, WolfeConst1, WolfeConst2, Increment)
        integer,intent(in)::objective,batch,dim
        type(c_ptr),intent(in)::x_dev,d_dev,b_dev,ws_dev,f_dev,iters_dev,status_dev
        integer(c_size_t),intent(in)::ws_bytes
        integer,intent(out)::info
        integer,intent(in),optional::Memory,MaxIteration
        logical,intent(in),optional::Strong
        real*8,intent(in),optional::Precision,MinStepLength,WolfeConst1,WolfeConst2,Increment
        type(fl_options)::o
        call fl_default_options(o,FL_SOLVER_LBFGS)
        if(present(Memory)) o%memory=Memory
        if(present(Strong)) o%strong=merge(1,0,Strong)
        if(present(MaxIteration)) o%max_iteration=MaxIteration
        if(present(Precision)) o%precision=Precision
        if(present(MinStepLength)) o%min_step_length=MinStepLength
        if(present(WolfeConst1)) o%wolfe_c1=WolfeConst1
        if(present(WolfeConst2)) o%wolfe_c2=WolfeConst2
        if(present(Increment)) o%increment=Increment
        info=fl_lbfgs_batched(objective,batch,dim,x_dev,d_dev,b_dev,o,ws_dev,ws_bytes,&
            f_dev,c_null_ptr,iters_dev,status_dev,c_null_ptr,c_null_ptr,c_null_ptr)
    end subroutine LBFGS_batched

    !Batched augmented Lagrangian (reference AugmentedLagrangian, NonlinearOptimization.f90:2005-2241) on device-resident
    !data: objective = built-in FL_OBJ_*, M block-sphere constraints, lambda_dev(M,batch) in (lambda0) / out; same keyword
    !names as AugmentedLagrangian.  ws_dev: fl_workspace_bytes_for(inner solver, ...) bytes.
    subroutine AugmentedLagrangian_batched(objective, x_dev, batch, N, M, d_dev, b_dev, lambda_dev, ws_dev, ws_bytes, &
    f_dev, cnorm2_dev, iters_dev, outer_dev, status_dev, info, &
    UnconstrainedSolver, miu0, ExactStep, Memory, Method, Strong, MaxIteration, Precision, MinStepLength, WolfeConst1, WolfeConst2, Increment)
        integer,intent(in)::objective,batch,N,M
        type(c_ptr),intent(in)::x_dev,d_dev,b_dev,lambda_dev,ws_dev,f_dev,cnorm2_dev,iters_dev,outer_dev,status_dev
        integer(c_size_t),intent(in)::ws_bytes
        integer,intent(out)::info
        character*(*),intent(in),optional::UnconstrainedSolver,Method
        real*8,intent(in),optional::miu0
        integer,intent(in),optional::ExactStep,Memory,MaxIteration
        logical,intent(in),optional::Strong
        real*8,intent(in),optional::Precision,MinStepLength,WolfeConst1,WolfeConst2,Increment
        type(fl_options)::o
        integer(c_int)::solver
        real(c_double)::mu
        solver=FL_SOLVER_BFGS!the reference's default (NonlinearOptimization.f90:2074)
        if(present(UnconstrainedSolver)) then
            select case(UnconstrainedSolver)
            case('NewtonRaphson'); solver=FL_SOLVER_NEWTON
            case('BFGS'); solver=FL_SOLVER_BFGS
            case('LBFGS'); solver=FL_SOLVER_LBFGS
            case('ConjugateGradient'); solver=FL_SOLVER_CG
            case default; stop 'Program abort: unsupported unconstrained solver'!NonlinearOptimization.f90:2186
            end select
        end if
        call fl_default_options(o,solver)
        if(present(Method)) then
            select case(Method)
            case('DY'); o%cg_method=0
            case('PR'); o%cg_method=1
            case default; stop 'Program abort: unsupported conjugate gradient method'
            end select
        end if
        if(present(ExactStep)) o%exact_step=ExactStep
        if(present(Memory)) o%memory=Memory
        if(present(Strong)) o%strong=merge(1,0,Strong)
        if(present(MaxIteration)) o%max_iteration=MaxIteration
        if(present(Precision)) o%precision=Precision
        if(present(MinStepLength)) o%min_step_length=MinStepLength
        if(present(WolfeConst1)) o%wolfe_c1=WolfeConst1
        if(present(WolfeConst2)) o%wolfe_c2=WolfeConst2
        if(present(Increment)) o%increment=Increment
        mu=1d0; if(present(miu0)) mu=miu0
        info=fl_augmented_lagrangian_batched(solver,objective,batch,N,M,x_dev,d_dev,b_dev,lambda_dev,mu,o,ws_dev,ws_bytes,&
            f_dev,cnorm2_dev,iters_dev,outer_dev,status_dev,c_null_ptr,c_null_ptr,c_null_ptr)
    end subroutine AugmentedLagrangian_batched
!------------------ End ------------------
end module NonlinearOptimization
