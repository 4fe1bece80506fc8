!Fortran caller of the BATCHED entry points (include/fl_nlopt.h through `use FortranLibrary`): what a Fortran program with a
!batch of problems uses instead of a loop over the one-problem routines.
!  1. AugmentedLagrangian_batched on device-resident data (n = 512, 8 block-sphere constraints: BASELINE config 5's shape),
!     held against fl_multi_solve on the same host arrays -- bit for bit.
!  2. A reverse-communication loop written in Fortran: the objective (the reference's callbacks f, fd:
!     NonlinearOptimization.f90:33-38) is evaluated HERE, on the host, for the whole batch between two fl_rci_step calls.
!  3. The objective handed over as HIP source text (fl_user_compile) and run inside the fused kernel -- equal to the
!     built-in diagonal quadratic bit for bit.
!Prints "Mission complete" when everything holds.  Device memory: the HIP runtime's C entry points, bound below.
module hip_bindings
    use iso_c_binding
    implicit none
    integer(c_int),parameter::hipMemcpyHostToDevice=1,hipMemcpyDeviceToHost=2
    interface
        integer(c_int) function hipMalloc(p,bytes) bind(C,name='hipMalloc')
            import; type(c_ptr),intent(out)::p; integer(c_size_t),value::bytes
        end function hipMalloc
        integer(c_int) function hipFree(p) bind(C,name='hipFree')
            import; type(c_ptr),value::p
        end function hipFree
        integer(c_int) function hipMemcpy(dst,src,bytes,kind) bind(C,name='hipMemcpy')
            import; type(c_ptr),value::dst,src; integer(c_size_t),value::bytes; integer(c_int),value::kind
        end function hipMemcpy
        integer(c_int) function hipMemset(p,v,bytes) bind(C,name='hipMemset')
            import; type(c_ptr),value::p; integer(c_int),value::v; integer(c_size_t),value::bytes
        end function hipMemset
        integer(c_int) function hipDeviceSynchronize() bind(C,name='hipDeviceSynchronize')
            import
        end function hipDeviceSynchronize
    end interface
end module hip_bindings

program main
    use FortranLibrary
    use hip_bindings
    implicit none
    integer,parameter::n=512,M=8,batch=48
    real(c_double),target::x(n,batch),d(n,batch),b(n,batch),x0(n,batch),lam(M,batch),f(batch),cn(batch)
    real(c_double),target::xm(n,batch),lamm(M,batch),fm(batch),cnm(batch),g(n,batch),xq(n,batch)
    integer(c_int),target::it(batch),ou(batch),st(batch),itm(batch),oum(batch),stm(batch),req(batch),nfm(batch),ngm(batch)
    type(c_ptr)::xd,dd,bd,ld,wsd,fd_,cnd,itd,oud,std,gd,rqd,h,uh
    type(fl_options)::o
    integer(c_size_t)::wsb,nb
    integer::i,k,info,rc,steps
    logical::ok
    character(len=4096),target::src
    character(kind=c_char),target::log(8192)
    ok=.true.
    do k=1,batch; do i=1,n
        d(i,k)=1d0+(1d0+dble(k)/batch*8d0)*dble(i-1)/dble(n-1)
        b(i,k)=sin(0.37d0*i+k)
        x0(i,k)=0.05d0+0.1d0*abs(cos(1.3d0*i+0.7d0*k))
    end do; end do
    nb=int(n,c_size_t)*batch*8
    call chk(hipMalloc(xd,nb)); call chk(hipMalloc(dd,nb)); call chk(hipMalloc(bd,nb)); call chk(hipMalloc(gd,nb))
    call chk(hipMalloc(ld,int(M*batch*8,c_size_t))); call chk(hipMalloc(fd_,int(batch*8,c_size_t))); call chk(hipMalloc(cnd,int(batch*8,c_size_t)))
    call chk(hipMalloc(itd,int(batch*4,c_size_t))); call chk(hipMalloc(oud,int(batch*4,c_size_t))); call chk(hipMalloc(std,int(batch*4,c_size_t)))
    call chk(hipMalloc(rqd,int(batch*4,c_size_t)))
    call chk(hipMemcpy(dd,c_loc(d),nb,hipMemcpyHostToDevice)); call chk(hipMemcpy(bd,c_loc(b),nb,hipMemcpyHostToDevice))

    write(*,*)'1. batched augmented Lagrangian on the device (L-BFGS inside), n = 512, M = 8'
    call chk(hipMemcpy(xd,c_loc(x0),nb,hipMemcpyHostToDevice)); call chk(hipMemset(ld,0,int(M*batch*8,c_size_t)))
    call fl_default_options(o,FL_SOLVER_LBFGS); o%precision=1d-9
    wsb=fl_workspace_bytes_for(FL_SOLVER_LBFGS,batch,n,o); call chk(hipMalloc(wsd,wsb))
    call AugmentedLagrangian_batched(FL_OBJ_DIAGQUAD,xd,batch,n,M,dd,bd,ld,wsd,wsb,fd_,cnd,itd,oud,std,info,&
        UnconstrainedSolver='LBFGS',Precision=1d-9)
    call chk(info); call chk(hipDeviceSynchronize())
    call chk(hipMemcpy(c_loc(x),xd,nb,hipMemcpyDeviceToHost)); call chk(hipMemcpy(c_loc(lam),ld,int(M*batch*8,c_size_t),hipMemcpyDeviceToHost))
    call chk(hipMemcpy(c_loc(cn),cnd,int(batch*8,c_size_t),hipMemcpyDeviceToHost)); call chk(hipMemcpy(c_loc(st),std,int(batch*4,c_size_t),hipMemcpyDeviceToHost))
    call chk(hipMemcpy(c_loc(ou),oud,int(batch*4,c_size_t),hipMemcpyDeviceToHost))
    write(*,'(A,ES12.4,A,I6)')' max |c| ',sqrt(maxval(cn)),'  outer iterations (max) ',maxval(ou)
    if(any(st/=0).or.sqrt(maxval(cn))>1d-9) then; write(*,*)'FAILED: constraints not met'; ok=.false.; end if
    !the same through fl_multi_solve (host arrays, every GPU of the node): bit for bit
    xm=x0; lamm=0d0
    rc=fl_multi_solve(FL_SOLVER_LBFGS,FL_OBJ_DIAGQUAD,batch,n,c_loc(xm),c_loc(d),c_loc(b),o,M,c_loc(lamm),1d0,c_loc(fm),c_null_ptr,&
        c_loc(cnm),c_loc(itm),c_loc(oum),c_loc(stm),c_loc(nfm),c_loc(ngm),0,0)
    call chk(rc)
    if(any(xm/=x).or.any(lamm/=lam).or.any(oum/=ou)) then; write(*,*)'FAILED: fl_multi_solve differs'; ok=.false.; end if
    write(*,'(A,L2)')' fl_multi_solve gives the same bits:',all(xm==x)

    write(*,*)'2. reverse communication from Fortran: f, f'' evaluated on the host between two steps (L-BFGS)'
    call fl_default_options(o,FL_SOLVER_LBFGS); o%precision=1d-8
    call chk(fl_rci_create(h,FL_SOLVER_LBFGS,batch,n,o,c_null_ptr))
    xq=0d0; call chk(hipMemcpy(xd,c_loc(xq),nb,hipMemcpyHostToDevice))
    call chk(fl_rci_step(h,xd,c_null_ptr,c_null_ptr,rqd))!first step: the initial guesses are the first points asked for
    steps=0
    do
        call chk(hipMemcpy(c_loc(req),rqd,int(batch*4,c_size_t),hipMemcpyDeviceToHost))
        if(all(req==0)) exit
        call chk(hipMemcpy(c_loc(xq),xd,nb,hipMemcpyDeviceToHost))
        do k=1,batch!the callbacks of the reference, for the problems that asked
            if(req(k)==0) cycle
            f(k)=0.5d0*sum(d(:,k)*xq(:,k)*xq(:,k))-sum(b(:,k)*xq(:,k))
            g(:,k)=d(:,k)*xq(:,k)-b(:,k)
        end do
        call chk(hipMemcpy(fd_,c_loc(f),int(batch*8,c_size_t),hipMemcpyHostToDevice)); call chk(hipMemcpy(gd,c_loc(g),nb,hipMemcpyHostToDevice))
        call chk(fl_rci_step(h,xd,fd_,gd,rqd))
        steps=steps+1
        if(steps>200000) then; write(*,*)'FAILED: no convergence'; ok=.false.; exit; end if
    end do
    call chk(hipMemcpy(c_loc(xq),xd,nb,hipMemcpyDeviceToHost)); call chk(fl_rci_results(h,fd_,c_null_ptr,itd,std,c_null_ptr,c_null_ptr))
    call chk(hipMemcpy(c_loc(st),std,int(batch*4,c_size_t),hipMemcpyDeviceToHost)); call chk(fl_rci_destroy(h))
    write(*,'(A,I7,A,ES12.4)')' steps ',steps,'  max |x - b/d| ',maxval(abs(xq-b/d))
    if(any(st>1).or.maxval(abs(xq-b/d))>1d-7) then; write(*,*)'FAILED: reverse communication'; ok=.false.; end if!(0 gradient / 1 step converged)

    write(*,*)'3. the objective as source text, compiled into the fused kernel at run time'
    src='template <int NW, int EPT> struct Quad {'//new_line('a')//&
        ' static constexpr int LDS_DOUBLES = 0; double d[EPT], b[EPT];'//new_line('a')//&
        ' __device__ void init(const fl::SolveArgs &A, int prob, double *) {'//new_line('a')//&
        '  fl::load_user<NW, EPT>(A.d + (size_t)prob * A.n, A.n, d); fl::load_user<NW, EPT>(A.b + (size_t)prob * A.n, A.n, b); }'//new_line('a')//&
        ' __device__ void eval(const double (&x)[EPT], double (&g)[EPT], double &s0, double &s1, int, double *) {'//new_line('a')//&
        '  for (int k = 0; k < EPT; ++k) { const double dx = d[k] * x[k]; const double t0 = dx * x[k], t1 = b[k] * x[k];'//new_line('a')//&
        '   g[k] = dx - b[k]; s0 = (k == 0) ? t0 : s0 + t0; s1 = (k == 0) ? t1 : s1 + t1; } }'//new_line('a')//&
        ' __device__ static double combine(double s0, double s1) { return 0.5 * s0 - s1; } };'//c_null_char
    rc=fl_user_compile(uh,src,'Quad'//c_null_char,FL_SOLVER_LBFGS,n,FL_OBJ_DIAGQUAD,log,int(size(log),c_size_t))
    if(rc/=0) then
        write(*,*)'FAILED: fl_user_compile ',rc; i=1
        do while(i<size(log).and.log(i)/=c_null_char); write(*,'(A)',advance='no')log(i); i=i+1; end do
        ok=.false.
    else
        call fl_default_options(o,FL_SOLVER_LBFGS); o%precision=1d-8
        xq=0d0; call chk(hipMemcpy(xd,c_loc(xq),nb,hipMemcpyHostToDevice))
        call chk(fl_user_solve(uh,batch,xd,dd,bd,c_null_ptr,o,wsd,wsb,fd_,c_null_ptr,itd,std,c_null_ptr,c_null_ptr,c_null_ptr))
        call chk(hipDeviceSynchronize()); call chk(hipMemcpy(c_loc(x),xd,nb,hipMemcpyDeviceToHost))
        xq=0d0; call chk(hipMemcpy(xd,c_loc(xq),nb,hipMemcpyHostToDevice))
        call LBFGS_batched(FL_OBJ_DIAGQUAD,xd,batch,n,dd,bd,wsd,wsb,fd_,itd,std,info,Precision=1d-8)
        call chk(info); call chk(hipDeviceSynchronize()); call chk(hipMemcpy(c_loc(xm),xd,nb,hipMemcpyDeviceToHost))
        write(*,'(A,L2,A,ES12.4)')' same bits as the built-in objective:',all(x==xm),'  max |x - b/d| ',maxval(abs(x-b/d))
        if(any(x/=xm).or.maxval(abs(x-b/d))>1d-7) then; write(*,*)'FAILED: compiled objective'; ok=.false.; end if
        call chk(fl_user_destroy(uh))
    end if
    call chk(hipFree(xd)); call chk(hipFree(dd)); call chk(hipFree(bd)); call chk(hipFree(gd)); call chk(hipFree(ld)); call chk(hipFree(wsd))
    if(ok) then; write(*,*)'Mission complete'; else; write(*,*)'FAILED'; stop 1; end if
contains
    subroutine chk(code)
        integer(c_int),intent(in)::code
        if(code/=0) then; write(*,*)'call failed with code ',code; stop 2; end if
    end subroutine chk
end program main
