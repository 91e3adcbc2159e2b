!> `module cls_obs_data` for the build's own Fortran host code: reads the opt_data.NNNNNN.dat files step 3
!> writes (n_sta rows x 7 list-directed columns: x y z t t_stdv amp amp_stdv, reference
!> src/cls_obs_data.f90:84-93) and offers the getters `type forward`'s constructor calls, plus the initial
!> x,y guess (station with the largest amplitude).  The reference also builds pairwise dt cubes that nothing
!> reads; they are not built here.
module cls_obs_data
  implicit none
  private
  public :: obs_data

  type obs_data
     integer :: n_events = 0, n_sta = 0
     double precision, allocatable :: t_obs(:,:), t_stdv(:,:), a_obs(:,:), a_stdv(:,:)   ! (n_sta, n_events)
     double precision, allocatable :: sta_x(:), sta_y(:)
   contains
     procedure :: make_initial_guess, get_t_obs, get_t_stdv, get_a_obs, get_a_stdv
  end type obs_data

  interface obs_data
     module procedure read_obs_data
  end interface obs_data

contains

  type(obs_data) function read_obs_data(win_id, n_sta, sta_x, sta_y, verb) result(o)
    integer, intent(in) :: win_id(:), n_sta
    double precision, intent(in) :: sta_x(:), sta_y(:)
    logical, intent(in) :: verb
    character(64) :: fname
    double precision :: skip(3)
    integer :: i, j, unit, ierr

    o%n_events = size(win_id)
    o%n_sta = n_sta
    allocate(o%t_obs(n_sta, o%n_events), o%t_stdv(n_sta, o%n_events))
    allocate(o%a_obs(n_sta, o%n_events), o%a_stdv(n_sta, o%n_events))
    o%sta_x = sta_x
    o%sta_y = sta_y
    if (verb) print *, "<< Reading obs files>>"
    do i = 1, o%n_events
       write(fname, '(A,I6.6,A)') "opt_data.", win_id(i), ".dat"
       open(newunit=unit, file=trim(fname), status="old", action="read", iostat=ierr)
       if (ierr /= 0) then
          print *, trim(fname)
          error stop "ERROR: obs_file is not found"
       end if
       do j = 1, n_sta
          read(unit, *) skip, o%t_obs(j, i), o%t_stdv(j, i), o%a_obs(j, i), o%a_stdv(j, i)
       end do
       close(unit)
    end do
  end function read_obs_data

  subroutine make_initial_guess(self, x_mu, y_mu)
    class(obs_data), intent(inout) :: self
    double precision, intent(out) :: x_mu(self%n_events), y_mu(self%n_events)
    integer :: i, k
    do i = 1, self%n_events
       k = maxloc(self%a_obs(:, i), dim=1)
       x_mu(i) = self%sta_x(k)
       y_mu(i) = self%sta_y(k)
    end do
  end subroutine make_initial_guess

  function get_t_obs(self) result(a)
    class(obs_data), intent(in) :: self
    double precision :: a(self%n_sta, self%n_events)
    a = self%t_obs
  end function get_t_obs

  function get_t_stdv(self) result(a)
    class(obs_data), intent(in) :: self
    double precision :: a(self%n_sta, self%n_events)
    a = self%t_stdv
  end function get_t_stdv

  function get_a_obs(self) result(a)
    class(obs_data), intent(in) :: self
    double precision :: a(self%n_sta, self%n_events)
    a = self%a_obs
  end function get_a_obs

  function get_a_stdv(self) result(a)
    class(obs_data), intent(in) :: self
    double precision :: a(self%n_sta, self%n_events)
    a = self%a_stdv
  end function get_a_stdv

end module cls_obs_data
