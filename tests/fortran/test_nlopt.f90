!Fortran smoke test of the drop-in module: every public routine of `use FortranLibrary` called the way the
!reference's own test program calls them (test/test.f90:330-478: all solvers, with and without f_fd / fdd, both line
!searches, the augmented Lagrangian on the unit sphere; "correct routines should print close to 0"), on an objective
!of this repository: the anharmonic bowl  f(x) = sum_i ( i x_i^2 / 2 + x_i^4 / 4 ),  minimum at x = 0.
!Deterministic start x_i = 0.1 i (the reference seeds random_number from the clock).
module bowl
    implicit none
contains
    subroutine f(fx,x,dim)
        real*8,intent(out)::fx
        integer,intent(in)::dim
        real*8,dimension(dim),intent(in)::x
        fx=sum(weights(dim)*x*x)/2d0+sum(x**4)/4d0
    end subroutine f
    subroutine fd(fdx,x,dim)
        integer,intent(in)::dim
        real*8,dimension(dim),intent(out)::fdx
        real*8,dimension(dim),intent(in)::x
        fdx=weights(dim)*x+x**3
    end subroutine fd
    integer function f_fd(fx,fdx,x,dim)
        integer,intent(in)::dim
        real*8,intent(out)::fx
        real*8,dimension(dim),intent(out)::fdx
        real*8,dimension(dim),intent(in)::x
        call f(fx,x,dim)
        call fd(fdx,x,dim)
        f_fd=0
    end function f_fd
    integer function fdd(fddx,x,N)
        integer,intent(in)::N
        real*8,dimension(N,N),intent(out)::fddx
        real*8,dimension(N),intent(in)::x
        real*8,dimension(N)::w
        integer::i
        w=weights(N)
        fddx=0d0
        forall(i=1:N) fddx(i,i)=w(i)+3d0*x(i)*x(i)
        fdd=0
    end function fdd
    function weights(N)
        integer,intent(in)::N
        real*8,dimension(N)::weights
        integer::i
        weights=[(dble(i),i=1,N)]
    end function weights
    !one equality constraint, the unit sphere: c(x) = x.x - 1
    subroutine c(cx,x,M,N)
        integer,intent(in)::M,N
        real*8,dimension(M),intent(out)::cx
        real*8,dimension(N),intent(in)::x
        cx(1)=dot_product(x,x)-1d0
    end subroutine c
    subroutine cd(cdx,x,M,N)
        integer,intent(in)::M,N
        real*8,dimension(N,M),intent(out)::cdx
        real*8,dimension(N),intent(in)::x
        cdx(:,1)=2d0*x
    end subroutine cd
end module bowl

program main
    use FortranLibrary
    use bowl
    implicit none
    integer,parameter::dim=10
    integer::i
    real*8,dimension(dim)::x,g,pdir
    real*8::fx0,fx1,a
    write(*,*)'Steepest descent'
    call start(); call SteepestDescent(f,fd,x,dim,Warning=.false.,MaxIteration=300); write(*,'(A,ES24.16)')' SD ',norm2(x)
    write(*,*)'Conjugate gradient'
    call start(); call ConjugateGradient(f,fd,x,dim,Warning=.false.); write(*,'(A,ES24.16)')' CG-DY ',norm2(x)
    call start(); call ConjugateGradient(f,fd,x,dim,Strong=.false.,Warning=.false.); write(*,'(A,ES24.16)')' CG-DY-Wolfe ',norm2(x)
    call start(); call ConjugateGradient(f,fd,x,dim,f_fd=f_fd,Warning=.false.); write(*,'(A,ES24.16)')' CG-DY-f_fd ',norm2(x)
    call start(); call ConjugateGradient(f,fd,x,dim,Method='PR',Warning=.false.); write(*,'(A,ES24.16)')' CG-PR ',norm2(x)
    write(*,*)'L-BFGS'
    call start(); call LBFGS(f,fd,x,dim,Warning=.false.); write(*,'(A,ES24.16)')' LBFGS ',norm2(x)
    call start(); call LBFGS(f,fd,x,dim,Strong=.true.,Warning=.false.); write(*,'(A,ES24.16)')' LBFGS-Strong ',norm2(x)
    call start(); call LBFGS(f,fd,x,dim,f_fd=f_fd,Memory=5,Warning=.false.); write(*,'(A,ES24.16)')' LBFGS-f_fd-M5 ',norm2(x)
    write(*,*)'BFGS'
    call start(); call BFGS(f,fd,x,dim,ExactStep=0,Warning=.false.); write(*,'(A,ES24.16)')' BFGS0 ',norm2(x)
    call start(); call BFGS(f,fd,x,dim,ExactStep=0,f_fd=f_fd,Warning=.false.); write(*,'(A,ES24.16)')' BFGS0-f_fd ',norm2(x)
    call start(); call BFGS(f,fd,x,dim,fdd=fdd,ExactStep=5,Warning=.false.); write(*,'(A,ES24.16)')' BFGS5-fdd ',norm2(x)
    write(*,*)'Newton'
    call start(); call NewtonRaphson(f,fd,x,dim,fdd=fdd,Warning=.false.); write(*,'(A,ES24.16)')' Newton ',norm2(x)
    call start(); call NewtonRaphson(f,fd,x,dim,fdd=fdd,f_fd=f_fd,Strong=.false.,Warning=.false.); write(*,'(A,ES24.16)')' Newton-f_fd-Wolfe ',norm2(x)
    write(*,*)'Newton-Raphson / BFGS with numerical Hessian (no fdd)'
    call start(); call NewtonRaphson(f,fd,x,dim,Warning=.false.); write(*,'(A,ES24.16)')' Newton-numH ',norm2(x)
    call start(); call BFGS(f,fd,x,dim,Warning=.false.); write(*,'(A,ES24.16)')' BFGS-default-numH ',norm2(x)
    write(*,*)'Augmented Lagrangian (unit sphere): | |x| - 1 |'
    call start(); call AugmentedLagrangian(f,fd,c,cd,x,dim,1,UnconstrainedSolver='LBFGS',Warning=.false.,&
        MaxIteration=100,Precision=1d-10); write(*,'(A,ES24.16)')' AugLag-LBFGS ',abs(norm2(x)-1d0)
    call start(); call AugmentedLagrangian(f,fd,c,cd,x,dim,1,UnconstrainedSolver='ConjugateGradient',Method='PR',f_fd=f_fd,&
        Warning=.false.,MaxIteration=100,Precision=1d-10); write(*,'(A,ES24.16)')' AugLag-CG-PR ',abs(norm2(x)-1d0)
    write(*,*)'Line searchers along -f''(x): f must decrease'
    call start(); call f(fx0,x,dim); call fd(g,x,dim); pdir=-g; fx1=fx0; a=1d0
    call StrongWolfe(1d-4,0.9d0,f,fd,x,a,pdir,fx1,-dot_product(g,g),g,dim); write(*,'(A,ES24.16)')' StrongWolfe-f/f0 ',fx1/fx0
    call start(); call fd(g,x,dim); pdir=-g; fx1=fx0; a=1d0
    call Wolfe_fdwithf(1d-4,0.9d0,f,fd,f_fd,x,a,pdir,fx1,-dot_product(g,g),g,dim,Increment=1.5d0)
    write(*,'(A,ES24.16)')' Wolfe-f/f0 ',fx1/fx0
    write(*,*)'Mission complete'
contains
    subroutine start()
        do i=1,dim; x(i)=0.1d0*i; end do
    end subroutine start
end program main
