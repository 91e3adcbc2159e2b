!> hypo_tremor_statistics on the MI355X: same command line, parameter file, input files (the step-5 sample files
!> of every rank, selected_win.dat, the station file) and output files as the reference's step-6 program
!> (src/hypo_tremor_statistics.f90, src/cls_statistics.f90), with the order statistics -- the reference quick-sorts
!> every parameter's samples -- taken by libhtm_hip.so's radix select on the device (htm_quantiles).
!>
!>   hypo_tremor_statistics_hip <parameter file>
!>
!> One process reads the files of all ranks (the reference gathers them onto rank 0 over MPI, :310-338).
program hypo_tremor_statistics_hip
  use, intrinsic :: iso_c_binding
  use, intrinsic :: iso_fortran_env, only: iostat_end
  use htm_c_api
  use htm_param, only: param, line_max, sample_convert
  implicit none

  type(param) :: para
  character(line_max) :: param_file
  integer, allocatable :: win_id(:)
  integer :: n_mod, n_evt, n_sta, io, ierr, id, device, ios
  integer(c_int) :: ranks(3)
  double precision :: dummy
  character(32) :: env
  real(c_double), allocatable :: smp(:,:), q_vs(:,:), q_qs(:,:), q_tc(:,:), q_ac(:,:), q_h(:,:)

  if (command_argument_count() /= 1) error stop "USAGE: hypo_tremor_statistics [parameter file]"
  call get_command_argument(1, param_file)
  call para%load(trim(param_file), verb=.true.)
  device = 0
  call get_environment_variable("HTM_DEVICE", env, status=ios)
  if (ios == 0) read(env, *, iostat=ios) device

  ! src/cls_statistics.f90:64 and :229-231 (single-precision products, truncated)
  n_mod = (para%n_iter - para%n_burn) * para%n_procs * para%n_cool / para%n_interval
  ranks(1) = int(0.025 * n_mod, c_int); ranks(2) = int(0.5 * n_mod, c_int); ranks(3) = int(0.975 * n_mod, c_int)
  if (ranks(1) < 1) error stop "n_mod < 40: the 2.5 % element would be element 0 of the sorted samples"

  allocate(win_id(0))
  open(newunit=io, file="selected_win.dat", status="old", action="read", iostat=ierr)
  if (ierr /= 0) error stop "ERROR: selected_win.dat is not found."
  do
     read(io, *, iostat=ierr) id, dummy
     if (ierr /= 0) exit
     win_id = [win_id, id]
  end do
  close(io)
  n_evt = size(win_id)
  n_sta = para%n_stations

  call quantiles_of("vs", 1, q_vs)
  call quantiles_of("qs", 1, q_qs)
  call write_vs_qs()
  call quantiles_of("t_corr", n_sta, q_tc)
  call quantiles_of("a_corr", n_sta, q_ac)
  call write_corr()
  call quantiles_of("hypo", 3 * n_evt, q_h)
  call write_hypo()

contains

  !> all ranks' records of one sample file family, stacked in rank order, and their (il, im, iu) elements
  subroutine quantiles_of(name, n_val, q)
    character(*), intent(in) :: name
    integer, intent(in) :: n_val
    real(c_double), allocatable, intent(out) :: q(:,:)
    character(64) :: f
    integer :: r, u, it, k, rc
    real(c_double) :: row(n_val)
    allocate(smp(n_val, n_mod), q(3, n_val))
    k = 0
    do r = 0, para%n_procs - 1
       write(f, '(A,A,I2.2,A)') name, ".", r, ".out"
       write(*, '(A)') "<< Now reading " // trim(f) // " >>"
       open(newunit=u, file=trim(f), status="old", access="stream", form="unformatted", convert=sample_convert(), iostat=ierr)
       if (ierr /= 0) then
          write(0, *) "ERROR: cannot open ", trim(f)
          error stop
       end if
       do
          read(u, iostat=ierr) it, row
          if (ierr /= 0) exit
          k = k + 1
          if (k > n_mod) error stop "more samples on file than n_mod = (n_iter - n_burn) * n_procs * n_cool / n_interval"
          smp(:, k) = row
       end do
       close(u)
    end do
    if (k /= n_mod) error stop "fewer samples on file than n_mod = (n_iter - n_burn) * n_procs * n_cool / n_interval"
    ! smp(n_val, n_mod) column-major = [n_mod][n_val] row-major: one recorded model per row
    rc = htm_quantiles(int(device, c_int), smp, int(n_mod, c_long), int(n_val, c_long), ranks, q)
    if (rc /= 0) then
       write(0, '(2A)') "ERROR: htm_quantiles: ", htm_error_message()
       error stop "libhtm_hip call failed"
    end if
    deallocate(smp)
  end subroutine quantiles_of

  subroutine write_vs_qs()
    integer :: u
    open(newunit=u, file="uniform_structure.stat", status="replace", form="formatted")
    write(u, '(A)') "# Vs (50%), Vs (2.5%) " // "Vs (97.5%), Qs (50%), Qs (2.5%), Qs (97.5%)"
    write(u, '(6F13.6)') q_vs(2, 1), q_vs(1, 1), q_vs(3, 1), q_qs(2, 1), q_qs(1, 1), q_qs(3, 1)
    close(u)
  end subroutine write_vs_qs

  subroutine write_corr()
    integer :: u, i
    open(newunit=u, file="station_corrections.stat", status="replace", form="formatted")
    write(u, '(A)') "# station name, t_corr (50%), t_corr (2.5%) " // &
         & "t_corr (97.5%), a_corr (50%), a_corr (2.5%), a_corr (97.5%)"
    do i = 1, n_sta
       write(u, '(A12,6F13.6)') trim(para%stations(i)), q_tc(2, i), q_tc(1, i), q_tc(3, i), q_ac(2, i), q_ac(1, i), q_ac(3, i)
    end do
    close(u)
  end subroutine write_corr

  !> hypo.stat, then hypo.stat.removed from the values as printed (the reference re-reads its own file, :136-142)
  subroutine write_hypo()
    integer :: u, i, j, ii, c
    integer, allocatable :: keep(:), new(:)
    double precision :: v(9, n_evt), lo, hi
    logical :: flag, inside
    character(13) :: cell
    open(newunit=u, file="hypo.stat", status="replace", form="formatted")
    write(u, '(A)') hypo_header()
    do i = 1, n_evt
       do c = 1, 3
          v(3*c-2, i) = q_h(2, 3*(i-1)+c); v(3*c-1, i) = q_h(1, 3*(i-1)+c); v(3*c, i) = q_h(3, 3*(i-1)+c)
       end do
       write(u, '(I9,9F13.6)') win_id(i), v(:, i)
    end do
    close(u)
    do i = 1, n_evt            ! what a reader of hypo.stat sees
       do c = 1, 9
          write(cell, '(F13.6)') v(c, i)
          read(cell, *) v(c, i)
       end do
    end do
    keep = [(i, i = 1, n_evt)]
    do                          ! src/cls_statistics.f90:150-185
       new = [keep(1)]
       flag = .false.
       do j = 2, size(keep)
          i = keep(j); ii = keep(j - 1)
          inside = .false.
          if (win_id(i) == win_id(ii) + 1) then
             inside = .true.
             do c = 1, 3
                lo = max(v(3*c-1, i), v(3*c-1, ii)); hi = min(v(3*c, i), v(3*c, ii))
                inside = inside .and. lo < v(3*c-2, i) .and. lo < v(3*c-2, ii) .and. hi > v(3*c-2, i) .and. hi > v(3*c-2, ii)
             end do
          end if
          if (inside) then
             flag = .true.
          else
             new = [new, i]
          end if
       end do
       keep = new
       if (.not. flag) exit
    end do
    open(newunit=u, file="hypo.stat.removed", status="replace", form="formatted")
    write(u, '(A)') hypo_header()
    do j = 1, size(keep)
       i = keep(j)
       write(u, '(I9,9F13.6)') win_id(i), (q_of(c, i), c = 1, 9)
    end do
    close(u)
  end subroutine write_hypo

  double precision function q_of(c, i)
    integer, intent(in) :: c, i
    integer :: cmp, which
    cmp = (c - 1) / 3 + 1; which = mod(c - 1, 3)      ! 0 median, 1 low, 2 high
    q_of = q_h(merge(2, merge(1, 3, which == 1), which == 0), 3*(i-1)+cmp)
  end function q_of

  function hypo_header() result(h)
    character(:), allocatable :: h
    h = "# window ID, x (50%), x (2.5%) " // "x (97.5%), y (50%), y (2.5%), y (97.5%)" // &
         & "z (50 %), z (2.5%), z (97.5%)"
  end function hypo_header

end program hypo_tremor_statistics_hip
