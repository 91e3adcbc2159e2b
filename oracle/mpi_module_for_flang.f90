! NOT reference code.  The conda MPICH in this image ships an `mpi.mod` compiled by
! gfortran, which AMD flang cannot read.  This file re-exposes MPICH's OWN header
! (/opt/conda/include/mpif.h) as a module so that the reference's `use mpi`
! (src/mod_mpi.f90:2) resolves against the installed MPI library.  Nothing is stubbed:
! all MPI symbols come from /opt/conda/lib/libmpifort.so + libmpi.so.
module mpi
  implicit none
  include 'mpif.h'
end module mpi
