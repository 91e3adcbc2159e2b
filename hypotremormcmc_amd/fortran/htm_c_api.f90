!> ISO_C_BINDING view of libhtm_hip.so (C ABI: include/htm_hip.h).  One interface per C symbol the Fortran
!> host code uses; argument names, order and meaning follow the header.
module htm_c_api
  use, intrinsic :: iso_c_binding
  implicit none
  private
  public :: htm_model_init, htm_chains_init
  public :: htm_last_error_c, htm_error_message
  public :: htm_forward_create, htm_forward_destroy, htm_forward_loglik_full, htm_forward_loglik_partial
  public :: htm_forward_travel_time, htm_forward_amp, htm_forward_travel_time_single, htm_forward_amp_single
  public :: htm_chains_create, htm_chains_destroy, htm_chains_run, htm_chains_get_state, htm_chains_get_rng, htm_chains_get_loglik, htm_chains_share_gpu
  public :: htm_chains_lik_count, htm_chains_lik_read, htm_chains_sample_count, htm_chains_sample_read
  public :: htm_chains_iterations_done
  public :: htm_chains_step_begin, htm_chains_swap_record_host, htm_chains_step_end_host, htm_chains_drain
  public :: htm_chains_xchg_handle, htm_chains_xchg_connect, htm_chains_xchg_probe, htm_chains_run_lockstep_direct
  public :: HTM_XCHG_HANDLE_BYTES, HTM_COMM_ID_BYTES
  public :: htm_comm_unique_id, htm_comm_create, htm_comm_destroy, htm_chains_run_lockstep_comm
  public :: htm_device_count, htm_device_physical_id, htm_quantiles, htm_select_regress
  public :: htm_chains_checkpoint_size, htm_chains_checkpoint_save, htm_chains_checkpoint_load

  integer(c_size_t), parameter :: HTM_XCHG_HANDLE_BYTES = 64_c_size_t, HTM_COMM_ID_BYTES = 128_c_size_t

  !> one `type model` group stacked over the chains of the rank (include/htm_hip.h: htm_model_init)
  type, bind(C) :: htm_model_init
     type(c_ptr) :: x = c_null_ptr, mu = c_null_ptr, sigma = c_null_ptr, step_size = c_null_ptr
     type(c_ptr) :: prior_type = c_null_ptr
  end type htm_model_init

  type, bind(C) :: htm_chains_init
     integer(c_int) :: n_chains, n_procs, rank
     type(htm_model_init) :: hypo, t_corr, vs, a_corr, qs
     type(c_ptr) :: temp
     integer(c_int) :: solve_vs, solve_t_corr, solve_qs, solve_a_corr
     integer(c_int32_t) :: rng_state(4)
     integer(c_int) :: n_burn, n_interval
     integer(c_int) :: lik_capacity = 0, sample_capacity = 0
  end type htm_chains_init

  interface
     function htm_last_error_c() bind(C, name="htm_last_error") result(p)
       import :: c_ptr
       type(c_ptr) :: p
     end function htm_last_error_c

     function htm_forward_create(n_sta, n_events, sta_x, sta_y, sta_z, t_obs, t_stdv, a_obs, a_stdv, &
          & use_time, use_amp, device, handle) bind(C, name="htm_forward_create") result(rc)
       import :: c_int, c_double, c_ptr
       integer(c_int), value :: n_sta, n_events, use_time, use_amp, device
       real(c_double), intent(in) :: sta_x(*), sta_y(*), sta_z(*)
       real(c_double), intent(in) :: t_obs(*), t_stdv(*), a_obs(*), a_stdv(*)
       type(c_ptr), intent(out) :: handle
       integer(c_int) :: rc
     end function htm_forward_create

     function htm_forward_destroy(handle) bind(C, name="htm_forward_destroy") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: handle
       integer(c_int) :: rc
     end function htm_forward_destroy

     function htm_forward_loglik_full(handle, hypo, t_corr, vs, a_corr, qs, log_likelihood) &
          & bind(C, name="htm_forward_loglik_full") result(rc)
       import :: c_int, c_double, c_ptr
       type(c_ptr), value :: handle
       real(c_double), intent(in) :: hypo(*), t_corr(*), a_corr(*)
       real(c_double), value :: vs, qs
       real(c_double), intent(out) :: log_likelihood
       integer(c_int) :: rc
     end function htm_forward_loglik_full

     function htm_forward_loglik_partial(handle, evt_id, hypo_old_xyz, log_likelihood_old, hypo_xyz, &
          & t_corr, vs, a_corr, qs, log_likelihood) bind(C, name="htm_forward_loglik_partial") result(rc)
       import :: c_int, c_double, c_ptr
       type(c_ptr), value :: handle
       integer(c_int), value :: evt_id
       real(c_double), intent(in) :: hypo_old_xyz(3), hypo_xyz(3), t_corr(*), a_corr(*)
       real(c_double), value :: log_likelihood_old, vs, qs
       real(c_double), intent(out) :: log_likelihood
       integer(c_int) :: rc
     end function htm_forward_loglik_partial

     function htm_forward_travel_time(handle, hypo, t_corr, vs, t_syn) &
          & bind(C, name="htm_forward_travel_time") result(rc)
       import :: c_int, c_double, c_ptr
       type(c_ptr), value :: handle
       real(c_double), intent(in) :: hypo(*), t_corr(*)
       real(c_double), value :: vs
       real(c_double), intent(out) :: t_syn(*)
       integer(c_int) :: rc
     end function htm_forward_travel_time

     function htm_forward_amp(handle, hypo, a_corr, qs, vs, a_syn) bind(C, name="htm_forward_amp") result(rc)
       import :: c_int, c_double, c_ptr
       type(c_ptr), value :: handle
       real(c_double), intent(in) :: hypo(*), a_corr(*)
       real(c_double), value :: qs, vs
       real(c_double), intent(out) :: a_syn(*)
       integer(c_int) :: rc
     end function htm_forward_amp

     function htm_forward_travel_time_single(handle, evt_id, hypo, t_corr, vs, t_syn) &
          & bind(C, name="htm_forward_travel_time_single") result(rc)
       import :: c_int, c_double, c_ptr
       type(c_ptr), value :: handle
       integer(c_int), value :: evt_id
       real(c_double), intent(in) :: hypo(*), t_corr(*)
       real(c_double), value :: vs
       real(c_double), intent(out) :: t_syn(*)
       integer(c_int) :: rc
     end function htm_forward_travel_time_single

     function htm_forward_amp_single(handle, evt_id, hypo, a_corr, qs, vs, a_syn) &
          & bind(C, name="htm_forward_amp_single") result(rc)
       import :: c_int, c_double, c_ptr
       type(c_ptr), value :: handle
       integer(c_int), value :: evt_id
       real(c_double), intent(in) :: hypo(*), a_corr(*)
       real(c_double), value :: qs, vs
       real(c_double), intent(out) :: a_syn(*)
       integer(c_int) :: rc
     end function htm_forward_amp_single

     function htm_chains_create(forward_handle, init, handle) bind(C, name="htm_chains_create") result(rc)
       import :: c_int, c_ptr, htm_chains_init
       type(c_ptr), value :: forward_handle
       type(htm_chains_init), intent(in) :: init
       type(c_ptr), intent(out) :: handle
       integer(c_int) :: rc
     end function htm_chains_create

     function htm_chains_destroy(handle) bind(C, name="htm_chains_destroy") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: handle
       integer(c_int) :: rc
     end function htm_chains_destroy

     function htm_chains_run(handle, n_iter) bind(C, name="htm_chains_run") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: handle
       integer(c_int), value :: n_iter
       integer(c_int) :: rc
     end function htm_chains_run

     function htm_chains_iterations_done(handle, n) bind(C, name="htm_chains_iterations_done") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: handle
       integer(c_int), intent(out) :: n
       integer(c_int) :: rc
     end function htm_chains_iterations_done

     !> lock-step iteration of a multi-rank job, records staged through host memory (MPI programs)
     function htm_chains_step_begin(handle) bind(C, name="htm_chains_step_begin") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: handle
       integer(c_int) :: rc
     end function htm_chains_step_begin
     function htm_chains_swap_record_host(handle, record) bind(C, name="htm_chains_swap_record_host") result(rc)
       import :: c_int, c_double, c_ptr
       type(c_ptr), value :: handle
       real(c_double), intent(out) :: record(*)
       integer(c_int) :: rc
     end function htm_chains_swap_record_host
     function htm_chains_step_end_host(handle, gathered) bind(C, name="htm_chains_step_end_host") result(rc)
       import :: c_int, c_double, c_ptr
       type(c_ptr), value :: handle
       real(c_double), intent(in) :: gathered(*)
       integer(c_int) :: rc
     end function htm_chains_step_end_host
     !> persistent lock-step (include/htm_hip.h): swap records exchanged inside the kernel through peer-mapped inboxes
     function htm_chains_xchg_handle(handle, ipc_handle, handle_bytes) bind(C, name="htm_chains_xchg_handle") result(rc)
       import :: c_int, c_ptr, c_char, c_size_t
       type(c_ptr), value :: handle
       character(kind=c_char), intent(out) :: ipc_handle(*)
       integer(c_size_t), value :: handle_bytes
       integer(c_int) :: rc
     end function htm_chains_xchg_handle
     function htm_chains_xchg_connect(handle, ipc_handles, handle_bytes) bind(C, name="htm_chains_xchg_connect") result(rc)
       import :: c_int, c_ptr, c_char, c_size_t
       type(c_ptr), value :: handle
       character(kind=c_char), intent(in) :: ipc_handles(*)
       integer(c_size_t), value :: handle_bytes
       integer(c_int) :: rc
     end function htm_chains_xchg_connect
     !> RCCL from Fortran (include/htm_hip.h): rank 0 draws the id, MPI_Bcast carries it, every rank creates its communicator
     function htm_comm_unique_id(id, id_bytes) bind(C, name="htm_comm_unique_id") result(rc)
       import :: c_int, c_char, c_size_t
       character(kind=c_char), intent(out) :: id(*)
       integer(c_size_t), value :: id_bytes
       integer(c_int) :: rc
     end function htm_comm_unique_id
     function htm_comm_create(id, id_bytes, rank, n_ranks, device, comm) bind(C, name="htm_comm_create") result(rc)
       import :: c_int, c_char, c_size_t, c_ptr
       character(kind=c_char), intent(in) :: id(*)
       integer(c_size_t), value :: id_bytes
       integer(c_int), value :: rank, n_ranks, device
       type(c_ptr), intent(out) :: comm
       integer(c_int) :: rc
     end function htm_comm_create
     function htm_comm_destroy(comm) bind(C, name="htm_comm_destroy") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: comm
       integer(c_int) :: rc
     end function htm_comm_destroy
     function htm_chains_run_lockstep_comm(handle, n_iter, comm) bind(C, name="htm_chains_run_lockstep_comm") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: handle, comm
       integer(c_int), value :: n_iter
       integer(c_int) :: rc
     end function htm_chains_run_lockstep_comm
     function htm_chains_xchg_probe(handle, token, seconds) bind(C, name="htm_chains_xchg_probe") result(rc)
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: handle
       integer(c_int), value :: token
       real(c_double), value :: seconds
       integer(c_int) :: rc
     end function htm_chains_xchg_probe
     function htm_chains_run_lockstep_direct(handle, n_iter) bind(C, name="htm_chains_run_lockstep_direct") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: handle
       integer(c_int), value :: n_iter
       integer(c_int) :: rc
     end function htm_chains_run_lockstep_direct
     function htm_chains_drain(handle) bind(C, name="htm_chains_drain") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: handle
       integer(c_int) :: rc
     end function htm_chains_drain
     !> step-6 order statistics (include/htm_hip.h): samples [n_mod][n_par] row-major, out [n_par][3]
     function htm_quantiles(device, samples, n_mod, n_par, ranks_1based, out) bind(C, name="htm_quantiles") result(rc)
       import :: c_int, c_long, c_double
       integer(c_int), value :: device
       real(c_double), intent(in) :: samples(*)
       integer(c_long), value :: n_mod, n_par
       integer(c_int), intent(in) :: ranks_1based(3)
       real(c_double), intent(out) :: out(*)
       integer(c_int) :: rc
     end function htm_quantiles
     !> step-4 regressions (include/htm_hip.h): t, t_err, a, a_err (n_sta, n_win); out (6, n_win) = vs, b, t0, a0, cc_t, cc_a
     function htm_select_regress(device, n_sta, n_win, sta_x, sta_y, sta_z, z_guess, t, t_err, a, a_err, out) &
          & bind(C, name="htm_select_regress") result(rc)
       import :: c_int, c_double
       integer(c_int), value :: device, n_sta, n_win
       real(c_double), intent(in) :: sta_x(*), sta_y(*), sta_z(*), t(*), t_err(*), a(*), a_err(*)
       real(c_double), value :: z_guess
       real(c_double), intent(out) :: out(*)
       integer(c_int) :: rc
     end function htm_select_regress
     function htm_device_count(n) bind(C, name="htm_device_count") result(rc)
       import :: c_int
       integer(c_int), intent(out) :: n
       integer(c_int) :: rc
     end function htm_device_count
     function htm_device_physical_id(device, id) bind(C, name="htm_device_physical_id") result(rc)
       import :: c_int
       integer(c_int), value :: device
       integer(c_int), intent(out) :: id
       integer(c_int) :: rc
     end function htm_device_physical_id

     !> checkpoint / resume (include/htm_hip.h); blob = c_loc of a byte buffer of htm_chains_checkpoint_size bytes
     function htm_chains_checkpoint_size(handle, bytes) bind(C, name="htm_chains_checkpoint_size") result(rc)
       import :: c_int, c_size_t, c_ptr
       type(c_ptr), value :: handle
       integer(c_size_t), intent(out) :: bytes
       integer(c_int) :: rc
     end function htm_chains_checkpoint_size
     function htm_chains_checkpoint_save(handle, blob, bytes) bind(C, name="htm_chains_checkpoint_save") result(rc)
       import :: c_int, c_size_t, c_ptr
       type(c_ptr), value :: handle, blob
       integer(c_size_t), value :: bytes
       integer(c_int) :: rc
     end function htm_chains_checkpoint_save
     function htm_chains_checkpoint_load(handle, blob, bytes) bind(C, name="htm_chains_checkpoint_load") result(rc)
       import :: c_int, c_size_t, c_ptr
       type(c_ptr), value :: handle, blob
       integer(c_size_t), value :: bytes
       integer(c_int) :: rc
     end function htm_chains_checkpoint_load

     function htm_chains_get_state(handle, chain, hypo, t_corr, vs, a_corr, qs, temp, log_likelihood, &
          & n_propose, n_accept) bind(C, name="htm_chains_get_state") result(rc)
       import :: c_int, c_int32_t, c_double, c_ptr
       type(c_ptr), value :: handle
       integer(c_int), value :: chain
       real(c_double), intent(out) :: hypo(*), t_corr(*), a_corr(*), vs, qs, temp, log_likelihood
       integer(c_int32_t), intent(out) :: n_propose(7), n_accept(7)
       integer(c_int) :: rc
     end function htm_chains_get_state

     function htm_chains_share_gpu(handle, ranks_on_this_gpu) bind(C, name="htm_chains_share_gpu") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: handle
       integer(c_int), value :: ranks_on_this_gpu
       integer(c_int) :: rc
     end function htm_chains_share_gpu

     function htm_chains_get_loglik(handle, chain, log_likelihood) bind(C, name="htm_chains_get_loglik") result(rc)
       import :: c_int, c_double, c_ptr
       type(c_ptr), value :: handle
       integer(c_int), value :: chain
       real(c_double), intent(out) :: log_likelihood
       integer(c_int) :: rc
     end function htm_chains_get_loglik

     function htm_chains_get_rng(handle, state) bind(C, name="htm_chains_get_rng") result(rc)
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr), value :: handle
       integer(c_int32_t), intent(out) :: state(4)
       integer(c_int) :: rc
     end function htm_chains_get_rng

     function htm_chains_lik_count(handle, n) bind(C, name="htm_chains_lik_count") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: handle
       integer(c_int), intent(out) :: n
       integer(c_int) :: rc
     end function htm_chains_lik_count

     function htm_chains_lik_read(handle, iter, chain, log_likelihood) bind(C, name="htm_chains_lik_read") result(rc)
       import :: c_int, c_int32_t, c_double, c_ptr
       type(c_ptr), value :: handle
       integer(c_int32_t), intent(out) :: iter(*), chain(*)
       real(c_double), intent(out) :: log_likelihood(*)
       integer(c_int) :: rc
     end function htm_chains_lik_read

     function htm_chains_sample_count(handle, n) bind(C, name="htm_chains_sample_count") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: handle
       integer(c_int), intent(out) :: n
       integer(c_int) :: rc
     end function htm_chains_sample_count

     function htm_chains_sample_read(handle, k, iter, chain, vs, qs, hypo, t_corr, a_corr) &
          & bind(C, name="htm_chains_sample_read") result(rc)
       import :: c_int, c_int32_t, c_double, c_ptr
       type(c_ptr), value :: handle
       integer(c_int), value :: k
       integer(c_int32_t), intent(out) :: iter, chain
       real(c_double), intent(out) :: vs, qs, hypo(*), t_corr(*), a_corr(*)
       integer(c_int) :: rc
     end function htm_chains_sample_read
  end interface

contains

  !> htm_last_error() as a Fortran string
  function htm_error_message() result(msg)
    character(len=:), allocatable :: msg
    type(c_ptr) :: p
    character(kind=c_char), pointer :: s(:)
    integer :: n
    p = htm_last_error_c()
    msg = ""
    if (.not. c_associated(p)) return
    call c_f_pointer(p, s, [512])
    n = 0
    do while (n < 512)
       if (s(n + 1) == c_null_char) exit
       n = n + 1
    end do
    allocate(character(len=n) :: msg)
    msg = transfer(s(1:n), msg)
  end function htm_error_message

end module htm_c_api
