!> hypo_tremor_mcmc on the MI355X: same command line, parameter file, input files and output files as the
!> reference's step-5 program (src/hypo_tremor_mcmc.f90), with the whole main loop -- propose, forward,
!> judge, record, swap (:236-284) -- executed by the device-resident chains of libhtm_hip.so.
!>
!>   hypo_tremor_mcmc_hip <parameter file>
!>
!> One process drives one GPU.  Built twice from this file (preprocessed): `hypo_tremor_mcmc_hip` (one rank, the
!> whole loop in one persistent kernel) and, with -DHTM_MPI, `hypo_tremor_mcmc_hip_mpi`: an MPI program like the
!> reference (`mpiexec -np N`, n_procs = N), rank r on GPU (r mod #GPUs) unless HTM_DEVICE says otherwise; the
!> ranks exchange one small record per iteration with MPI_Allgather (the reference: one MPI_Bcast + two
!> send/recv pairs, src/cls_parallel.f90:118-213) -- the RCCL form of the same exchange is the Python launcher's,
!> see INTEGRATION.md.  Set-up follows the reference line by line so that the random stream is consumed in the
!> same order: station corrections, amplitude corrections, hypocentres, then the temperature of each tempered chain.
program hypo_tremor_mcmc_hip
#ifdef HTM_MPI
  use mpi
#endif
  use, intrinsic :: iso_c_binding
  use, intrinsic :: iso_fortran_env, only: iostat_end
  use htm_c_api
  use htm_random
  use htm_param, only: param, line_max, sample_convert
  use cls_model, only: model
  use cls_obs_data, only: obs_data
  use cls_forward, only: forward, htm_default_device
  implicit none

  type(param) :: para
  type(obs_data) :: obs
  type(forward) :: fwd
  type(model) :: t_corr, a_corr, hypo, vs, qs
  type(htm_chains_init) :: init
  type(c_ptr) :: chains
  character(line_max) :: param_file
  integer, allocatable :: win_id(:)
  double precision, allocatable :: x_mu(:), y_mu(:)
  integer :: n_events, n_sta, n_chains, i, j, io, id, ierr, rank, n_ranks
  integer :: slice = 10000          ! iterations per call into the library = cadence of the progress report (the reference reports every 1000, src/cls_mcmc.f90:230: HTM_SLICE=1000; measured 2.6 % of the main loop at that cadence, 0.3 % at 10000)
  integer(8) :: t_loop0 = 0
  logical :: direct = .false., want_rccl = .false.
  integer(c_int) :: n_dev
  real(c_double), allocatable :: rec(:), gathered(:)
  double precision :: dummy
  double precision, parameter :: eps = epsilon(1.d0)
  ! chain-major stacks handed to htm_chains_create
  real(c_double), allocatable, target :: hx(:,:), hmu(:,:), hsg(:,:), hst(:,:)
  real(c_double), allocatable, target :: tx(:,:), tmu(:,:), tsg(:,:), tst(:,:)
  real(c_double), allocatable, target :: ax(:,:), amu(:,:), asg(:,:), ast(:,:)
  real(c_double), allocatable, target :: vx(:), vmu(:), vsg(:), vst(:), qx(:), qmu(:), qsg(:), qst(:), temps(:)
  integer(c_int32_t), allocatable, target :: hpt(:,:), tpt(:,:), apt(:,:), vpt(:), qpt(:)

  rank = 0; n_ranks = 1
#ifdef HTM_MPI
  call mpi_init(ierr)
  call mpi_comm_size(MPI_COMM_WORLD, n_ranks, ierr)
  call mpi_comm_rank(MPI_COMM_WORLD, rank, ierr)
  if (htm_device_count(n_dev) == 0 .and. n_dev > 0) htm_default_device = mod(rank, int(n_dev))
#endif
  if (command_argument_count() /= 1) error stop "USAGE: hypo_tremor_mcmc [parameter file]"
  call get_command_argument(1, param_file)
  call para%load(trim(param_file), verb=(rank == 0))
  if (para%n_procs /= n_ranks) then
     if (rank == 0) then
        write(*, *) "ERROR: n_procs in parameter file must be equal to that is given in the command line"
#ifndef HTM_MPI
        write(*, *) "       (this program drives one GPU; hypo_tremor_mcmc_hip_mpi or the Python launcher run n_procs > 1)"
#endif
     end if
#ifdef HTM_MPI
     call mpi_abort(MPI_COMM_WORLD, MPI_ERR_OTHER, ierr)
#endif
     stop 1
  end if
  call rng_seed([5551111, 453222, 4444431, 6765], rank)
  block
    character(16) :: ev
    integer :: el, es, v
    call get_environment_variable("HTM_SLICE", ev, el, es)
    if (es == 0 .and. el > 0) then
       read(ev(1:el), *, iostat=es) v
       if (es == 0 .and. v >= 1) slice = v
    end if
  end block

  ! events
  allocate(win_id(0))
  open(newunit=io, file="selected_win.dat", status="old", action="read", iostat=ierr)
  if (ierr /= 0) error stop "cannot open selected_win.dat"
  do
     read(io, *, iostat=ierr) id, dummy
     if (ierr == iostat_end) exit
     win_id = [win_id, id]
  end do
  close(io)
  n_events = size(win_id)
  n_sta = para%n_stations
  n_chains = para%n_chains

  obs = obs_data(win_id=win_id, n_sta=n_sta, sta_x=para%sta_x, sta_y=para%sta_y, verb=(rank == 0))
  allocate(x_mu(n_events), y_mu(n_events))
  call obs%make_initial_guess(x_mu, y_mu)
  fwd = forward(n_sta=n_sta, n_events=n_events, sta_x=para%sta_x, sta_y=para%sta_y, sta_z=para%sta_z, &
       & obs=obs, use_amp=para%use_amp, use_time=para%use_time)

  allocate(hx(3*n_events, n_chains), hmu(3*n_events, n_chains), hsg(3*n_events, n_chains), hst(3*n_events, n_chains))
  allocate(hpt(3*n_events, n_chains))
  allocate(tx(n_sta, n_chains), tmu(n_sta, n_chains), tsg(n_sta, n_chains), tst(n_sta, n_chains), tpt(n_sta, n_chains))
  allocate(ax(n_sta, n_chains), amu(n_sta, n_chains), asg(n_sta, n_chains), ast(n_sta, n_chains), apt(n_sta, n_chains))
  allocate(vx(n_chains), vmu(n_chains), vsg(n_chains), vst(n_chains), vpt(n_chains))
  allocate(qx(n_chains), qmu(n_chains), qsg(n_chains), qst(n_chains), qpt(n_chains), temps(n_chains))

  do j = 1, n_chains
     t_corr = model(nx=n_sta)
     if (para%solve_t_corr) then
        do i = 1, n_sta
           call t_corr%set_prior(i=i, mu=para%prior_t_corr, sigma=para%prior_width_t_corr)
           call t_corr%set_perturb(i=i, step_size=para%step_size_t_corr)
        end do
        call t_corr%generate_model()
     else
        t_corr%x = para%prior_t_corr
     end if
     a_corr = model(nx=n_sta)
     if (para%solve_a_corr) then
        do i = 1, n_sta
           call a_corr%set_prior(i=i, mu=para%prior_a_corr, sigma=para%prior_width_a_corr)
           call a_corr%set_perturb(i=i, step_size=para%step_size_a_corr)
        end do
        call a_corr%generate_model()
     else
        a_corr%x = para%prior_a_corr
     end if
     hypo = model(nx=3*n_events)
     do i = 1, n_events
        call hypo%set_prior(i=3*i-2, mu=x_mu(i), sigma=para%prior_width_xy)
        call hypo%set_prior(i=3*i-1, mu=y_mu(i), sigma=para%prior_width_xy)
        call hypo%set_prior(i=3*i, mu=para%prior_z, sigma=para%prior_width_z, prior_type=1)
        call hypo%set_perturb(i=3*i-2, step_size=para%step_size_xy)
        call hypo%set_perturb(i=3*i-1, step_size=para%step_size_xy)
        call hypo%set_perturb(i=3*i, step_size=para%step_size_z)
     end do
     call hypo%generate_model()
     vs = model(nx=1)
     call vs%set_prior(i=1, mu=para%prior_vs, sigma=para%prior_width_vs)
     call vs%set_perturb(i=1, step_size=para%step_size_vs)
     call vs%set_x(1, para%prior_vs)
     qs = model(nx=1)
     call qs%set_prior(i=1, mu=para%prior_qs, sigma=para%prior_width_qs)
     call qs%set_perturb(i=1, step_size=para%step_size_qs)
     call qs%set_x(1, para%prior_qs)

     hx(:, j) = hypo%x;  hmu(:, j) = hypo%mu;  hsg(:, j) = hypo%sigma;  hst(:, j) = hypo%step_size
     hpt(:, j) = int(hypo%prior_type, c_int32_t)
     tx(:, j) = t_corr%x; tmu(:, j) = t_corr%mu; tsg(:, j) = t_corr%sigma; tst(:, j) = t_corr%step_size
     tpt(:, j) = int(t_corr%prior_type, c_int32_t)
     ax(:, j) = a_corr%x; amu(:, j) = a_corr%mu; asg(:, j) = a_corr%sigma; ast(:, j) = a_corr%step_size
     apt(:, j) = int(a_corr%prior_type, c_int32_t)
     vx(j) = vs%x(1); vmu(j) = vs%mu(1); vsg(j) = vs%sigma(1); vst(j) = vs%step_size(1); vpt(j) = 0
     qx(j) = qs%x(1); qmu(j) = qs%mu(1); qsg(j) = qs%sigma(1); qst(j) = qs%step_size(1); qpt(j) = 0
     if (j <= para%n_cool) then
        temps(j) = 1.d0
     else
        temps(j) = exp((rand_u() * (1.d0 - eps) + eps) * log(para%temp_high))
     end if
  end do

  init%n_chains = n_chains; init%n_procs = n_ranks; init%rank = rank
  init%hypo = pack_model(hx, hmu, hsg, hst, hpt)
  init%t_corr = pack_model(tx, tmu, tsg, tst, tpt)
  init%a_corr = pack_model(ax, amu, asg, ast, apt)
  init%vs%x = c_loc(vx); init%vs%mu = c_loc(vmu); init%vs%sigma = c_loc(vsg); init%vs%step_size = c_loc(vst)
  init%vs%prior_type = c_loc(vpt)
  init%qs%x = c_loc(qx); init%qs%mu = c_loc(qmu); init%qs%sigma = c_loc(qsg); init%qs%step_size = c_loc(qst)
  init%qs%prior_type = c_loc(qpt)
  init%temp = c_loc(temps)
  init%solve_vs = merge(1, 0, para%solve_vs); init%solve_t_corr = merge(1, 0, para%solve_t_corr)
  init%solve_qs = merge(1, 0, para%solve_qs); init%solve_a_corr = merge(1, 0, para%solve_a_corr)
  init%rng_state = rng_state()
  init%n_burn = para%n_burn; init%n_interval = para%n_interval
  call check(htm_chains_create(fwd%c_handle(), init, chains), "htm_chains_create")

  if (rank == 0) print *, "start MCMC"
#ifdef HTM_MPI
  ! main loop of src/hypo_tremor_mcmc.f90:236-284 with swap_temperature (src/cls_parallel.f90:100-216) as ONE
  ! all-gather of every rank's record {pair chosen by rank 0, pending judge_swap draw, (T, L) of its chains}
  ! Ranks that share a GPU (fewer GPUs than ranks on a node): every rank's persistent launch must be resident at once, so each
  ! takes its share of the CUs.  The ranks of this node tell each other which device they use.
  block
    integer :: node_comm, n_node, k, same, my_gpu
    integer(c_int) :: pci
    integer, allocatable :: devs(:)
    call mpi_comm_split_type(MPI_COMM_WORLD, MPI_COMM_TYPE_SHARED, 0, MPI_INFO_NULL, node_comm, ierr)
    call mpi_comm_size(node_comm, n_node, ierr)
    allocate(devs(n_node))
    ! (the GPU by its PCI address, not by its ordinal: with a visible-devices mask per rank every rank's ordinal is 0)
    my_gpu = -1 - int(htm_default_device)
    if (htm_device_physical_id(int(max(0, htm_default_device), c_int), pci) == 0) my_gpu = int(pci)
    call mpi_allgather(my_gpu, 1, MPI_INTEGER, devs, 1, MPI_INTEGER, node_comm, ierr)
    same = 0
    do k = 1, n_node
       if (devs(k) == my_gpu) same = same + 1
    end do
    if (same > 1) call check(htm_chains_share_gpu(chains, int(same, c_int)), "htm_chains_share_gpu")
    call mpi_comm_free(node_comm, ierr)
  end block
  ! Fastest transport first: persistent lock-step -- every rank's kernel writes its record straight into the other
  ! ranks' inboxes (peer-mapped device memory, xGMI between the GPUs of a node) and never leaves the GPU.  Set-up =
  ! one MPI_Allgather of the inboxes' IPC handles; taken only if EVERY rank could map every peer (HTM_XCHG=0: never).
  block
    character(kind=c_char) :: xh(HTM_XCHG_HANDLE_BYTES)
    character(kind=c_char), allocatable :: xh_all(:)
    character(len=8) :: env
    integer :: ok_mine, ok_all, elen, estat
    allocate(xh_all(HTM_XCHG_HANDLE_BYTES * n_ranks))
    xh = c_null_char
    ok_mine = 1
    call get_environment_variable("HTM_XCHG", env, elen, estat)
    if (estat == 0 .and. elen > 0) then
       if (env(1:1) == "0") ok_mine = 0
       if (env(1:min(4, elen)) == "rccl") then      ! HTM_XCHG=rccl: one RCCL all-gather per iteration from this Fortran host
          ok_mine = 0
          want_rccl = .true.
       end if
    end if
    if (ok_mine == 1) then
       if (htm_chains_xchg_handle(chains, xh, HTM_XCHG_HANDLE_BYTES) /= 0) ok_mine = 0
    end if
    call mpi_allgather(xh, int(HTM_XCHG_HANDLE_BYTES), MPI_BYTE, xh_all, int(HTM_XCHG_HANDLE_BYTES), MPI_BYTE, &
         & MPI_COMM_WORLD, ierr)
    if (ok_mine == 1) then
       if (htm_chains_xchg_connect(chains, xh_all, HTM_XCHG_HANDLE_BYTES) /= 0) ok_mine = 0
    end if
    call mpi_allreduce(ok_mine, ok_all, 1, MPI_INTEGER, MPI_MIN, MPI_COMM_WORLD, ierr)
    if (ok_all == 1) then     ! every rank mapped every peer: prove that writes into the inboxes reach a polling kernel
       call mpi_barrier(MPI_COMM_WORLD, ierr)
       if (htm_chains_xchg_probe(chains, 1_c_int, 10.0_c_double) /= 0) ok_mine = 0
       call mpi_allreduce(ok_mine, ok_all, 1, MPI_INTEGER, MPI_MIN, MPI_COMM_WORLD, ierr)
    end if
    direct = ok_all == 1
  end block
  if (direct) then
     if (rank == 0) print *, "swap exchange: in-kernel (peer-mapped inboxes)"
     call mpi_barrier(MPI_COMM_WORLD, ierr)
     call loop_timer(.true.)
     do i = 0, para%n_iter - 1, slice
        call check(htm_chains_run_lockstep_direct(chains, int(min(slice, para%n_iter - i), c_int)), &
             & "htm_chains_run_lockstep_direct")
        if (rank == 0) call summary(min(i + slice, para%n_iter))
     end do
     call loop_timer(.false.)
  end if
  ! Second transport: RCCL driven from Fortran.  Rank 0 draws RCCL's unique id, MPI_Bcast carries it, every rank joins the
  ! communicator (ncclCommInitRank), and the loop of htm_chains_run_lockstep -- one k_mcmc launch + one ncclAllGather
  ! per iteration, all enqueued from C on the chains' stream -- runs without a host synchronisation per iteration.
  ! RCCL wants one device per rank (it refuses two ranks on one GPU): then the program falls through to the next path.
  if (.not. direct .and. want_rccl) then
     block
       character(kind=c_char) :: cid(HTM_COMM_ID_BYTES)
       type(c_ptr) :: comm
       integer :: ok_mine, ok_all
       cid = c_null_char
       ok_mine = 1
       if (rank == 0) then
          if (htm_comm_unique_id(cid, HTM_COMM_ID_BYTES) /= 0) ok_mine = 0
       end if
       call mpi_bcast(cid, int(HTM_COMM_ID_BYTES), MPI_BYTE, 0, MPI_COMM_WORLD, ierr)
       call mpi_allreduce(ok_mine, ok_all, 1, MPI_INTEGER, MPI_MIN, MPI_COMM_WORLD, ierr)
       if (ok_all == 1) then
          if (htm_comm_create(cid, HTM_COMM_ID_BYTES, int(rank, c_int), int(n_ranks, c_int), &
               & int(htm_default_device, c_int), comm) /= 0) ok_mine = 0
          call mpi_allreduce(ok_mine, ok_all, 1, MPI_INTEGER, MPI_MIN, MPI_COMM_WORLD, ierr)
       end if
       if (ok_all == 1) then
          if (rank == 0) print *, "swap exchange: RCCL all-gather per iteration"
          call loop_timer(.true.)
          do i = 0, para%n_iter - 1, slice
             call check(htm_chains_run_lockstep_comm(chains, int(min(slice, para%n_iter - i), c_int), comm), &
                  & "htm_chains_run_lockstep_comm")
             call check(htm_chains_drain(chains), "htm_chains_drain")
             if (rank == 0) call summary(min(i + slice, para%n_iter))
          end do
          call loop_timer(.false.)
          direct = .true.         ! (the iterations are done)
          call check(htm_comm_destroy(comm), "htm_comm_destroy")
       else if (rank == 0) then
          print *, "RCCL communicator not available: ", htm_error_message()
       end if
     end block
  end if
  allocate(rec(4 + 2 * n_chains), gathered((4 + 2 * n_chains) * n_ranks))
  do i = 1, merge(0, para%n_iter, direct)      ! compatibility path: records staged through host memory + MPI_Allgather
     call check(htm_chains_step_begin(chains), "htm_chains_step_begin")
     call check(htm_chains_swap_record_host(chains, rec), "htm_chains_swap_record_host")
     call mpi_allgather(rec, size(rec), MPI_DOUBLE_PRECISION, gathered, size(rec), MPI_DOUBLE_PRECISION, &
          & MPI_COMM_WORLD, ierr)
     call check(htm_chains_step_end_host(chains, gathered), "htm_chains_step_end_host")
     if (mod(i, 1000) == 0) then
        call check(htm_chains_drain(chains), "htm_chains_drain")
        if (rank == 0) call summary(i)
     end if
  end do
  call check(htm_chains_drain(chains), "htm_chains_drain")
#else
  call loop_timer(.true.)
  do i = 0, para%n_iter - 1, slice
     call check(htm_chains_run(chains, int(min(slice, para%n_iter - i), c_int)), "htm_chains_run")
     call summary(min(i + slice, para%n_iter))
  end do
  call loop_timer(.false.)
#endif

  call write_outputs()
  call check(htm_chains_destroy(chains), "htm_chains_destroy")
#ifdef HTM_MPI
  call mpi_finalize(ierr)
#endif

contains

  subroutine check(rc, where)
    integer(c_int), intent(in) :: rc
    character(*), intent(in) :: where
    if (rc /= 0) then
       write(0, '(4A)') "ERROR: ", where, ": ", htm_error_message()
       error stop "libhtm_hip call failed"
    end if
  end subroutine check

  function pack_model(x, mu, sg, st, pt) result(m)
    real(c_double), intent(in), target :: x(:,:), mu(:,:), sg(:,:), st(:,:)
    integer(c_int32_t), intent(in), target :: pt(:,:)
    type(htm_model_init) :: m
    m%x = c_loc(x); m%mu = c_loc(mu); m%sigma = c_loc(sg); m%step_size = c_loc(st); m%prior_type = c_loc(pt)
  end function pack_model

  !> HTM_TIME_MAIN_LOOP=1: wall time of the main loop alone (src/hypo_tremor_mcmc.f90:236-284) on standard error
  subroutine loop_timer(start)
    logical, intent(in) :: start
    character(8) :: ev
    integer :: el, es
    integer(8) :: now, rate
    real(8) :: sec
    call get_environment_variable("HTM_TIME_MAIN_LOOP", ev, el, es)
    if (es /= 0 .or. el < 1) return
    if (ev(1:1) == "0") return
    call system_clock(now, rate)
    if (start) then
       t_loop0 = now
    else if (rank == 0) then
       sec = real(now - t_loop0, 8) / real(rate, 8)
       write(0, '(A,F12.6,A,I0,A,I0,A,I0,A,F14.1,A)') "main loop: ", sec, " s for ", para%n_iter, " iterations x ", n_chains, &
            & " chains x ", n_ranks, " rank(s) = ", real(para%n_iter, 8) * n_chains * n_ranks / sec, " proposal steps/s"
    end if
  end subroutine loop_timer

  !> the reference prints chain 1 every 1000 iterations (src/cls_mcmc.f90:230-237); HTM_SLICE=n makes a slice of the
  !> main loop -- one call into the library, one progress report -- n iterations long instead
  subroutine summary(it)
    integer, intent(in) :: it
    real(c_double) :: l
    call check(htm_chains_get_loglik(chains, 0_c_int, l), "htm_chains_get_loglik")
    write(*, *) "Iteration    : ", it, "/", para%n_iter
    write(*, *) "Likelihood   : ", l
    write(*, *)
  end subroutine summary

  !> hypo.RR.out, t_corr.RR.out, vs.RR.out, a_corr.RR.out, qs.RR.out, likelihoodRR.out (stream access,
  !> unformatted, src/hypo_tremor_mcmc.f90:216-233,:270-280) and proposal_count.txt (src/cls_parallel.f90:270-278)
  subroutine write_outputs()
    integer(c_int) :: n, k
    integer(c_int32_t), allocatable :: it(:), ch(:)
    real(c_double), allocatable :: lk(:)
    integer(c_int32_t) :: it1, ch1, np(7), na(7), np_sum(7), na_sum(7)
    real(c_double) :: h(3*n_events), tc(n_sta), ac(n_sta), v, q, t, l
    integer :: io_h, io_t, io_v, io_a, io_q, io_l, c
    character(5), parameter :: label(7) = [character(5) :: "vs", "t_corr", "qs", "a_corr", "x", "y", "z"]
    character(32) :: f

    write(f, '(A,I2.2,A)') "likelihood", rank, ".out"
    open(newunit=io_l, file=trim(f), status="replace", access="stream", form="unformatted", convert=sample_convert())
    call check(htm_chains_lik_count(chains, n), "htm_chains_lik_count")
    allocate(it(n), ch(n), lk(n))
    if (n > 0) call check(htm_chains_lik_read(chains, it, ch, lk), "htm_chains_lik_read")
    do k = 1, n
       write(io_l) int(it(k)), lk(k)
    end do
    close(io_l)

    write(f, '(A,I2.2,A)') "hypo.", rank, ".out";   open(newunit=io_h, file=trim(f), status="replace", access="stream", form="unformatted", convert=sample_convert())
    write(f, '(A,I2.2,A)') "t_corr.", rank, ".out"; open(newunit=io_t, file=trim(f), status="replace", access="stream", form="unformatted", convert=sample_convert())
    write(f, '(A,I2.2,A)') "vs.", rank, ".out";     open(newunit=io_v, file=trim(f), status="replace", access="stream", form="unformatted", convert=sample_convert())
    write(f, '(A,I2.2,A)') "a_corr.", rank, ".out"; open(newunit=io_a, file=trim(f), status="replace", access="stream", form="unformatted", convert=sample_convert())
    write(f, '(A,I2.2,A)') "qs.", rank, ".out";     open(newunit=io_q, file=trim(f), status="replace", access="stream", form="unformatted", convert=sample_convert())
    call check(htm_chains_sample_count(chains, n), "htm_chains_sample_count")
    do k = 0, n - 1
       call check(htm_chains_sample_read(chains, k, it1, ch1, v, q, h, tc, ac), "htm_chains_sample_read")
       write(io_v) int(it1), v
       write(io_h) int(it1), h
       write(io_t) int(it1), tc
       write(io_q) int(it1), q
       write(io_a) int(it1), ac
    end do
    close(io_h); close(io_t); close(io_v); close(io_a); close(io_q)

    np_sum = 0; na_sum = 0
    do c = 0, n_chains - 1
       call check(htm_chains_get_state(chains, int(c, c_int), h, tc, v, ac, q, t, l, np, na), "htm_chains_get_state")
       np_sum = np_sum + np; na_sum = na_sum + na
    end do
#ifdef HTM_MPI
    ! src/cls_parallel.f90:244-281: counters of all ranks summed on rank 0
    block
      integer(c_int32_t) :: np_all(7), na_all(7)
      integer :: ie
      call mpi_reduce(np_sum, np_all, 7, MPI_INTEGER4, MPI_SUM, 0, MPI_COMM_WORLD, ie)
      call mpi_reduce(na_sum, na_all, 7, MPI_INTEGER4, MPI_SUM, 0, MPI_COMM_WORLD, ie)
      np_sum = np_all; na_sum = na_all
    end block
#endif
    if (rank == 0) then
       open(newunit=io_l, file="proposal_count.txt", status="unknown")
       do c = 1, 7
          write(io_l, '(A,2I10)') '"' // label(c) // '"', np_sum(c), na_sum(c)
       end do
       close(io_l)
    end if
  end subroutine write_outputs

end program hypo_tremor_mcmc_hip
