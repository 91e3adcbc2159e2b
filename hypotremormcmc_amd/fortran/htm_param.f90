!> Parameter-file and station-file reader for the step-5 driver, with the reference's grammar
!> (src/cls_line_text.f90:88-148, src/cls_param.f90:212-255,:294-346,:350-390,:429-541):
!>   `#` starts a comment; ALL blanks are removed; `name=value`; lines without `=` or with an empty side are
!>   skipped; an unknown name stops the program; every key the MCMC program requires must be given;
!>   prior_t_corr / prior_a_corr default to 0.  Keys of the other pipeline steps are accepted and ignored, so
!>   the reference's sample/hypo_tremor.in reads unchanged.
module htm_param
  implicit none
  private
  public :: param, line_max, sample_convert

  integer, parameter :: line_max = 200

  type param
     character(line_max) :: station_file = ""
     integer :: n_procs = 0, n_iter = 0, n_burn = 0, n_interval = 0, n_chains = 0, n_cool = 0
     double precision :: temp_high = 0.d0
     double precision :: prior_z = 0.d0, prior_width_z = 0.d0, prior_width_xy = 0.d0
     double precision :: prior_vs = 0.d0, prior_width_vs = 0.d0, prior_qs = 0.d0, prior_width_qs = 0.d0
     double precision :: prior_t_corr = 0.d0, prior_width_t_corr = 0.d0
     double precision :: prior_a_corr = 0.d0, prior_width_a_corr = 0.d0
     double precision :: step_size_z = 0.d0, step_size_xy = 0.d0, step_size_vs = 0.d0
     double precision :: step_size_qs = 0.d0, step_size_t_corr = 0.d0, step_size_a_corr = 0.d0
     logical :: solve_vs = .false., solve_t_corr = .false., solve_qs = .false., solve_a_corr = .false.
     logical :: use_time = .false., use_amp = .false.
     integer :: n_stations = 0
     character(line_max), allocatable :: stations(:)
     double precision, allocatable :: sta_x(:), sta_y(:), sta_z(:)
     character(32) :: given(100) = ""
     integer :: n_given = 0
     logical :: verb = .false.
   contains
     procedure :: load, read_station_file
  end type param

  character(32), parameter :: required_mcmc(29) = [character(32) :: &
       & "n_procs", "station_file", "n_iter", "n_burn", "n_interval", "n_chains", "n_cool", "temp_high", &
       & "prior_z", "prior_width_z", "prior_width_xy", "prior_vs", "prior_width_vs", "prior_qs", &
       & "prior_width_qs", "prior_width_t_corr", "prior_width_a_corr", "step_size_z", "step_size_xy", &
       & "step_size_vs", "step_size_qs", "step_size_t_corr", "step_size_a_corr", "solve_vs", &
       & "solve_t_corr", "solve_qs", "solve_a_corr", "use_time", "use_amp"]
  ! keys of pipeline steps 1-4: legal in a shared parameter file, unused here
  character(32), parameter :: other_keys(15) = [character(32) :: &
       & "time_id_file", "cmp1", "cmp2", "data_dir", "filename_format", "t_win_conv", "t_win_corr", &
       & "t_step_corr", "n_pair_thred", "alpha", "vs_min", "vs_max", "b_min", "b_max", "z_guess"]

contains

  !> CONVERT= value of the unformatted sample / likelihood files.  The reference's stock gfortran Makefile builds
  !> with -fconvert=big-endian (src/Makefile:7-9): a step 5 / step 6 built that way writes / expects big-endian
  !> records, any other build native little-endian.  HTM_SAMPLE_ENDIAN=big|little (default little).
  function sample_convert() result(cv)
    character(:), allocatable :: cv
    character(32) :: e
    integer :: n, st
    call get_environment_variable("HTM_SAMPLE_ENDIAN", e, n, st)
    if (st /= 0 .or. n == 0) then
       cv = "little_endian"
    else if (e(1:n) == "big" .or. e(1:n) == "big_endian" .or. e(1:n) == "big-endian") then
       cv = "big_endian"
    else if (e(1:n) == "little" .or. e(1:n) == "little_endian" .or. e(1:n) == "little-endian" .or. e(1:n) == "native") then
       cv = "little_endian"
    else
       write(0, *) "ERROR: HTM_SAMPLE_ENDIAN must be big or little, got ", e(1:n)
       error stop
    end if
  end function sample_convert

  subroutine load(self, file, verb)
    class(param), intent(inout) :: self
    character(*), intent(in) :: file
    logical, intent(in) :: verb
    character(line_max) :: line, name, val
    integer :: unit, ierr, k, n, j, i

    self%verb = verb
    open(newunit=unit, file=file, status="old", action="read", iostat=ierr)
    if (ierr /= 0) then
       if (verb) write(0, *) "ERROR: cannot open ", trim(file)
       stop
    end if
    do
       read(unit, '(a)', iostat=ierr) line
       if (ierr /= 0) exit
       k = index(line, "#")
       if (k > 0) line = line(:k - 1)
       n = 0                                       ! squeeze out every blank
       do k = 1, len_trim(line)
          if (line(k:k) /= " ") then
             n = n + 1
             line(n:n) = line(k:k)
          end if
       end do
       line(n + 1:) = ""
       j = index(line(:n), "=")
       if (j <= 1 .or. j == n) cycle
       name = line(:j - 1)
       val = line(j + 1:n)
       if (verb) write(*, *) trim(name), " <- ", trim(val)
       call assign(self, trim(name), trim(val))
       self%n_given = self%n_given + 1
       self%given(self%n_given) = name
    end do
    close(unit)
    do i = 1, size(required_mcmc)
       if (.not. any(self%given(:self%n_given) == required_mcmc(i))) then
          if (verb) write(*, *) "ERROR: ", trim(required_mcmc(i)), " is not given."
          stop
       end if
    end do
    call self%read_station_file()
  end subroutine load

  subroutine assign(self, name, val)
    class(param), intent(inout) :: self
    character(*), intent(in) :: name, val
    select case (name)
    case ("station_file");        self%station_file = val
    case ("n_procs");             read(val, *) self%n_procs
    case ("n_iter");              read(val, *) self%n_iter
    case ("n_burn");              read(val, *) self%n_burn
    case ("n_interval");          read(val, *) self%n_interval
    case ("n_chains");            read(val, *) self%n_chains
    case ("n_cool");              read(val, *) self%n_cool
    case ("temp_high");           read(val, *) self%temp_high
    case ("prior_z");             read(val, *) self%prior_z
    case ("prior_width_z");       read(val, *) self%prior_width_z
    case ("prior_width_xy");      read(val, *) self%prior_width_xy
    case ("prior_vs");            read(val, *) self%prior_vs
    case ("prior_width_vs");      read(val, *) self%prior_width_vs
    case ("prior_qs");            read(val, *) self%prior_qs
    case ("prior_width_qs");      read(val, *) self%prior_width_qs
    case ("prior_t_corr");        read(val, *) self%prior_t_corr
    case ("prior_width_t_corr");  read(val, *) self%prior_width_t_corr
    case ("prior_a_corr");        read(val, *) self%prior_a_corr
    case ("prior_width_a_corr");  read(val, *) self%prior_width_a_corr
    case ("step_size_z");         read(val, *) self%step_size_z
    case ("step_size_xy");        read(val, *) self%step_size_xy
    case ("step_size_vs");        read(val, *) self%step_size_vs
    case ("step_size_qs");        read(val, *) self%step_size_qs
    case ("step_size_t_corr");    read(val, *) self%step_size_t_corr
    case ("step_size_a_corr");    read(val, *) self%step_size_a_corr
    case ("solve_vs");            read(val, *) self%solve_vs
    case ("solve_t_corr");        read(val, *) self%solve_t_corr
    case ("solve_qs");            read(val, *) self%solve_qs
    case ("solve_a_corr");        read(val, *) self%solve_a_corr
    case ("use_time");            read(val, *) self%use_time
    case ("use_amp");             read(val, *) self%use_amp
    case default
       if (.not. any(other_keys == name)) then
          if (self%verb) then
             write(0, *) "ERROR: Invalid parameter name"
             write(0, *) "        : ", name, "  (?)"
          end if
          stop
       end if
    end select
  end subroutine assign

  !> 6 list-directed columns: name x y z amp_fac1 amp_fac2
  subroutine read_station_file(self)
    class(param), intent(inout) :: self
    integer :: unit, ierr, n, i
    double precision :: fac(2)
    open(newunit=unit, file=self%station_file, status="old", action="read", iostat=ierr)
    if (ierr /= 0) then
       write(0, *) "ERROR: cannot open", trim(self%station_file)
       stop
    end if
    n = 0
    do
       read(unit, *, iostat=ierr)
       if (ierr /= 0) exit
       n = n + 1
    end do
    self%n_stations = n
    allocate(self%stations(n), self%sta_x(n), self%sta_y(n), self%sta_z(n))
    rewind(unit)
    do i = 1, n
       read(unit, *) self%stations(i), self%sta_x(i), self%sta_y(i), self%sta_z(i), fac
       if (self%verb) write(*, '(1x,a,3F9.3)') trim(self%stations(i)), self%sta_x(i), self%sta_y(i), self%sta_z(i)
    end do
    close(unit)
  end subroutine read_station_file

end module htm_param
