!> Drop-in replacement of the reference's `module cls_forward` (src/cls_forward.f90): the same public type
!> name, constructor keywords and type-bound procedure names / argument order, implemented as a thin
!> ISO_C_BINDING shim over libhtm_hip.so (include/htm_hip.h).  A maintainer swaps src/cls_forward.f90 for
!> this file plus htm_c_api.f90 and links -lhtm_hip (INTEGRATION.md); `program main`
!> (src/hypo_tremor_mcmc.f90:101-110,:247-257) compiles unchanged against it.
!>
!> It `use`s the caller's own cls_obs_data / cls_model modules -- only obs%get_t_obs() .. get_a_stdv(),
!> model%get_x(i), model%get_nx() are needed, exactly what the reference implementation reads.
!> Errors: the reference returns none; a failing HIP call ends the program with `error stop` and the
!> library's message, which is how the reference reports its own fatal conditions (src/cls_obs_data.f90:88).
module cls_forward
  use, intrinsic :: iso_c_binding
  use cls_obs_data, only: obs_data
  use cls_model, only: model
  use htm_c_api
  implicit none
  private
  public :: forward, htm_default_device

  !> device used when HTM_DEVICE is not set; the MPI driver sets it to (local rank mod device count)
  integer, save :: htm_default_device = -1

  type forward
     private
     integer :: n_events = 0
     integer :: n_sta = 0
     type(c_ptr) :: handle = c_null_ptr
   contains
     procedure :: calc_log_likelihood => forward_calc_log_likelihood
     procedure :: partially_update_log_likelihood => forward_partially_update_log_likelihood
     procedure :: calc_travel_time => forward_calc_travel_time
     procedure :: calc_amp => forward_calc_amp
     procedure :: calc_travel_time_single => forward_calc_travel_time_single
     procedure :: calc_amp_single => forward_calc_amp_single
     procedure :: c_handle => forward_c_handle
  end type forward

  interface forward
     module procedure init_forward
  end interface forward

contains

  subroutine check(rc, where)
    integer(c_int), intent(in) :: rc
    character(*), intent(in) :: where
    if (rc /= 0) then
       write(0, '(4A)') "ERROR: ", where, ": ", htm_error_message()
       error stop "libhtm_hip call failed"
    end if
  end subroutine check

  !> values of a `type(model)` as a plain vector (the reference reads them one by one through get_x)
  function model_values(m) result(x)
    type(model), intent(in) :: m
    real(c_double), allocatable :: x(:)
    integer :: i
    allocate(x(m%get_nx()))
    do i = 1, size(x)
       x(i) = m%get_x(i)
    end do
  end function model_values

  type(forward) function init_forward(n_sta, n_events, sta_x, sta_y, sta_z, obs, use_amp, use_time) result(self)
    integer, intent(in) :: n_sta, n_events
    double precision, intent(in) :: sta_x(:), sta_y(:), sta_z(:)
    type(obs_data), intent(in) :: obs
    logical, intent(in) :: use_amp, use_time
    real(c_double), allocatable :: t_obs(:,:), t_stdv(:,:), a_obs(:,:), a_stdv(:,:)
    integer :: device, ios
    character(32) :: env

    self%n_sta = n_sta
    self%n_events = n_events
    t_obs = obs%get_t_obs()          ! (n_sta, n_events), column-major: what the C ABI expects
    t_stdv = obs%get_t_stdv()
    a_obs = obs%get_a_obs()
    a_stdv = obs%get_a_stdv()
    device = max(0, htm_default_device)   ! one rank per GPU: HTM_DEVICE, else what the MPI driver chose, else 0
    call get_environment_variable("HTM_DEVICE", env, status=ios)
    if (ios == 0) read(env, *, iostat=ios) device
    call check(htm_forward_create(int(n_sta, c_int), int(n_events, c_int), sta_x, sta_y, sta_z, &
         & t_obs, t_stdv, a_obs, a_stdv, merge(1_c_int, 0_c_int, use_time), merge(1_c_int, 0_c_int, use_amp), &
         & int(device, c_int), self%handle), "forward(...)")
  end function init_forward

  type(c_ptr) function forward_c_handle(self) result(h)
    class(forward), intent(in) :: self
    h = self%handle
  end function forward_c_handle

  subroutine forward_calc_log_likelihood(self, hypo, t_corr, vs, a_corr, qs, log_likelihood)
    class(forward), intent(inout) :: self
    type(model), intent(in) :: hypo, t_corr, vs, a_corr, qs
    double precision, intent(out) :: log_likelihood
    call check(htm_forward_loglik_full(self%handle, model_values(hypo), model_values(t_corr), vs%get_x(1), &
         & model_values(a_corr), qs%get_x(1), log_likelihood), "calc_log_likelihood")
  end subroutine forward_calc_log_likelihood

  subroutine forward_partially_update_log_likelihood(self, evt_id, hypo_old, log_likelihood_old, hypo, &
       & t_corr, vs, a_corr, qs, log_likelihood)
    class(forward), intent(inout) :: self
    integer, intent(in) :: evt_id
    type(model), intent(in) :: hypo_old, hypo, t_corr, vs, a_corr, qs
    double precision, intent(in) :: log_likelihood_old
    double precision, intent(out) :: log_likelihood
    real(c_double) :: xyz_old(3), xyz_new(3)
    integer :: k
    do k = 1, 3                       ! only event evt_id of the two models is read (cls_forward.f90:151-153)
       xyz_old(k) = hypo_old%get_x(3 * (evt_id - 1) + k)
       xyz_new(k) = hypo%get_x(3 * (evt_id - 1) + k)
    end do
    call check(htm_forward_loglik_partial(self%handle, int(evt_id, c_int), xyz_old, log_likelihood_old, xyz_new, &
         & model_values(t_corr), vs%get_x(1), model_values(a_corr), qs%get_x(1), log_likelihood), &
         & "partially_update_log_likelihood")
  end subroutine forward_partially_update_log_likelihood

  subroutine forward_calc_travel_time(self, hypo, t_corr, vs, t_syn)
    class(forward), intent(inout) :: self
    type(model), intent(in) :: hypo, t_corr, vs
    double precision, intent(out) :: t_syn(self%n_sta, self%n_events)
    call check(htm_forward_travel_time(self%handle, model_values(hypo), model_values(t_corr), vs%get_x(1), t_syn), &
         & "calc_travel_time")
  end subroutine forward_calc_travel_time

  subroutine forward_calc_amp(self, hypo, a_corr, qs, vs, a_syn)
    class(forward), intent(inout) :: self
    type(model), intent(in) :: hypo, a_corr, qs, vs
    double precision, intent(out) :: a_syn(self%n_sta, self%n_events)
    call check(htm_forward_amp(self%handle, model_values(hypo), model_values(a_corr), qs%get_x(1), vs%get_x(1), &
         & a_syn), "calc_amp")
  end subroutine forward_calc_amp

  subroutine forward_calc_travel_time_single(self, evt_id, hypo, t_corr, vs, t_syn)
    class(forward), intent(inout) :: self
    integer, intent(in) :: evt_id
    type(model), intent(in) :: hypo, t_corr, vs
    double precision, intent(out) :: t_syn(self%n_sta)
    call check(htm_forward_travel_time_single(self%handle, int(evt_id, c_int), model_values(hypo), &
         & model_values(t_corr), vs%get_x(1), t_syn), "calc_travel_time_single")
  end subroutine forward_calc_travel_time_single

  subroutine forward_calc_amp_single(self, evt_id, hypo, a_corr, qs, vs, a_syn)
    class(forward), intent(inout) :: self
    integer, intent(in) :: evt_id
    type(model), intent(in) :: hypo, a_corr, qs, vs
    double precision, intent(out) :: a_syn(self%n_sta)
    call check(htm_forward_amp_single(self%handle, int(evt_id, c_int), model_values(hypo), model_values(a_corr), &
         & qs%get_x(1), vs%get_x(1), a_syn), "calc_amp_single")
  end subroutine forward_calc_amp_single

end module cls_forward
