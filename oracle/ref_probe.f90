! ref_probe.f90 -- NOT reference code.  A small driver (ours) that links against the UNMODIFIED reference
! modules compiled into oracle/_ref (mod_random, cls_model, cls_obs_data, cls_forward) and dumps
! known-answer vectors for the hot path, so that the C restatement and the HIP kernels can be pinned
! against the reference's own arithmetic at a finer grain than whole-run traces.
!
! Run in a directory holding opt_data.NNNNNN.dat files (read through the reference's own reader,
! src/cls_obs_data.f90:74-116) and probe_in.txt (written by tests/golden/make_golden.py):
!   n_sta n_events use_time use_amp n_cases
!   sta_x(:) / sta_y(:) / sta_z(:)
!   per case: hypo(3E) / t_corr(S) / vs / a_corr(S) / qs / evt_id / new xyz(3)
! Writes probe_out.txt:
!   rng section, then per case: L_full, L_partial(evt_id: hypo -> hypo with event moved, L_old = L_full),
!   L_full(moved); for case 1 additionally t_syn(S,E), a_syn(S,E), t_syn_single(S), a_syn_single(S).
program ref_probe
  use mod_random
  use cls_model, only: model
  use cls_obs_data, only: obs_data
  use cls_forward, only: forward
  implicit none
  integer :: n_sta, n_events, n_cases, io, oo, i, j, k, evt_id, r
  logical :: use_time, use_amp
  integer, allocatable :: win_id(:)
  double precision, allocatable :: sta_x(:), sta_y(:), sta_z(:), h(:), tc(:), ac(:)
  double precision, allocatable :: t_syn(:,:), a_syn(:,:), t1(:), a1(:), x_mu(:), y_mu(:)
  double precision :: vs_v, qs_v, xyz(3), l_full, l_part, l_moved
  type(model) :: hypo, hypo2, t_corr, a_corr, vs, qs
  type(obs_data) :: obs
  type(forward) :: fwd

  open(newunit=io, file="probe_in.txt", status="old")
  open(newunit=oo, file="probe_out.txt", status="replace")
  read(io,*) n_sta, n_events, use_time, use_amp, n_cases
  allocate(sta_x(n_sta), sta_y(n_sta), sta_z(n_sta), win_id(n_events))
  allocate(h(3*n_events), tc(n_sta), ac(n_sta))
  allocate(t_syn(n_sta,n_events), a_syn(n_sta,n_events), t1(n_sta), a1(n_sta))
  allocate(x_mu(n_events), y_mu(n_events))
  read(io,*) sta_x
  read(io,*) sta_y
  read(io,*) sta_z
  do i = 1, n_events
     win_id(i) = i
  end do

  ! --- RNG golden vectors (src/mod_random.f90) ---
  do r = 0, 3
     call init_random(5551111, 453222, 4444431, 6765, r)
     write(oo,'(A,I2)') "rng rank", r
     do k = 1, 8
        write(oo,'(ES26.17E3)') rand_u()
     end do
     write(oo,'(ES26.17E3)') rand_u2()
     write(oo,'(ES26.17E3)') rand_g()
     write(oo,'(ES26.17E3)') rand_r()
     write(oo,'(ES26.17E3)') rand_g()
  end do

  obs = obs_data(win_id=win_id, n_sta=n_sta, sta_x=sta_x, sta_y=sta_y, verb=.false.)
  call obs%make_initial_guess(x_mu, y_mu)
  write(oo,'(A)') "initial_guess"
  do i = 1, n_events
     write(oo,'(2ES26.17E3)') x_mu(i), y_mu(i)
  end do
  fwd = forward(n_sta=n_sta, n_events=n_events, sta_x=sta_x, sta_y=sta_y, sta_z=sta_z, &
       & obs=obs, use_amp=use_amp, use_time=use_time)

  hypo = model(nx=3*n_events)
  hypo2 = model(nx=3*n_events)
  t_corr = model(nx=n_sta)
  a_corr = model(nx=n_sta)
  vs = model(nx=1)
  qs = model(nx=1)
  do k = 1, n_cases
     read(io,*) h
     read(io,*) tc
     read(io,*) vs_v
     read(io,*) ac
     read(io,*) qs_v
     read(io,*) evt_id
     read(io,*) xyz
     do i = 1, 3*n_events
        call hypo%set_x(i, h(i))
        call hypo2%set_x(i, h(i))
     end do
     do j = 1, 3
        call hypo2%set_x(3*(evt_id-1)+j, xyz(j))
     end do
     do j = 1, n_sta
        call t_corr%set_x(j, tc(j))
        call a_corr%set_x(j, ac(j))
     end do
     call vs%set_x(1, vs_v)
     call qs%set_x(1, qs_v)
     call fwd%calc_log_likelihood(hypo, t_corr, vs, a_corr, qs, l_full)
     call fwd%partially_update_log_likelihood(evt_id, hypo, l_full, hypo2, t_corr, vs, a_corr, qs, l_part)
     call fwd%calc_log_likelihood(hypo2, t_corr, vs, a_corr, qs, l_moved)
     write(oo,'(A,I4)') "case", k
     write(oo,'(3ES26.17E3)') l_full, l_part, l_moved
     if (k == 1) then
        call fwd%calc_travel_time(hypo, t_corr, vs, t_syn)
        call fwd%calc_amp(hypo, a_corr, qs, vs, a_syn)
        call fwd%calc_travel_time_single(evt_id, hypo, t_corr, vs, t1)
        call fwd%calc_amp_single(evt_id, hypo, a_corr, qs, vs, a1)
        write(oo,'(A)') "t_syn"
        do i = 1, n_events
           do j = 1, n_sta
              write(oo,'(ES26.17E3)') t_syn(j,i)
           end do
        end do
        write(oo,'(A)') "a_syn"
        do i = 1, n_events
           do j = 1, n_sta
              write(oo,'(ES26.17E3)') a_syn(j,i)
           end do
        end do
        write(oo,'(A)') "t_syn_single"
        do j = 1, n_sta
           write(oo,'(ES26.17E3)') t1(j)
        end do
        write(oo,'(A)') "a_syn_single"
        do j = 1, n_sta
           write(oo,'(ES26.17E3)') a1(j)
        end do
     end if
  end do
  close(io)
  close(oo)
end program ref_probe
