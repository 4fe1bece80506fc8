!Umbrella module: `use FortranLibrary` (reference: source/FortranLibrary.f90:2-17 re-exports its ten
!modules; this build provides the one on the MI355X hot path)
module FortranLibrary
    use NonlinearOptimization
    implicit none
end module FortranLibrary
