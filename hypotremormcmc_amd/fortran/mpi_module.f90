!> `module mpi` for compilers that cannot read the MPI installation's own mpi.mod (the image's conda MPICH was
!> built with gfortran, the build uses AMD flang): MPICH's Fortran-77 header as a module.  Every MPI symbol comes
!> from the installed libmpifort / libmpi.
module mpi
  implicit none
  include 'mpif.h'
end module mpi
