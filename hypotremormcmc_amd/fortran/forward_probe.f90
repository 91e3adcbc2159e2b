!> Calls the drop-in `type forward` (cls_forward_hip.f90) the way the reference's main program does and prints
!> the results, for the parity test tests/test_gpu_fortran.py.  Same probe_in.txt format as the oracle's
!> reference probe (oracle/ref_probe.f90); run in a directory with opt_data.NNNNNN.dat files.
program forward_probe
  use cls_model, only: model
  use cls_obs_data, only: obs_data
  use cls_forward, only: forward
  implicit none
  integer :: n_sta, n_events, n_cases, io, oo, i, j, k, evt_id
  logical :: use_time, use_amp
  integer, allocatable :: win_id(:)
  double precision, allocatable :: sta_x(:), sta_y(:), sta_z(:), h(:), tc(:), ac(:), t1(:), a1(:)
  double precision :: vs_v, qs_v, xyz(3), l_full, l_part, l_moved
  type(model) :: hypo, hypo2, t_corr, a_corr, vs, qs
  type(obs_data) :: obs
  type(forward) :: fwd

  open(newunit=io, file="probe_in.txt", status="old")
  open(newunit=oo, file="probe_out_hip.txt", status="replace")
  read(io, *) n_sta, n_events, use_time, use_amp, n_cases
  allocate(sta_x(n_sta), sta_y(n_sta), sta_z(n_sta), win_id(n_events), h(3*n_events), tc(n_sta), ac(n_sta))
  allocate(t1(n_sta), a1(n_sta))
  read(io, *) sta_x
  read(io, *) sta_y
  read(io, *) sta_z
  win_id = [(i, i = 1, n_events)]
  obs = obs_data(win_id=win_id, n_sta=n_sta, sta_x=sta_x, sta_y=sta_y, verb=.false.)
  fwd = forward(n_sta=n_sta, n_events=n_events, sta_x=sta_x, sta_y=sta_y, sta_z=sta_z, obs=obs, &
       & use_amp=use_amp, use_time=use_time)
  hypo = model(nx=3*n_events); hypo2 = model(nx=3*n_events)
  t_corr = model(nx=n_sta); a_corr = model(nx=n_sta); vs = model(nx=1); qs = model(nx=1)
  do k = 1, n_cases
     read(io, *) h
     read(io, *) tc
     read(io, *) vs_v
     read(io, *) ac
     read(io, *) qs_v
     read(io, *) evt_id
     read(io, *) xyz
     hypo%x = h; hypo2%x = h
     hypo2%x(3*(evt_id-1)+1:3*evt_id) = xyz
     t_corr%x = tc; a_corr%x = ac
     call vs%set_x(1, vs_v)
     call qs%set_x(1, qs_v)
     call fwd%calc_log_likelihood(hypo, t_corr, vs, a_corr, qs, l_full)
     call fwd%partially_update_log_likelihood(evt_id, hypo, l_full, hypo2, t_corr, vs, a_corr, qs, l_part)
     call fwd%calc_log_likelihood(hypo2, t_corr, vs, a_corr, qs, l_moved)
     write(oo, '(3ES26.17E3)') l_full, l_part, l_moved
     if (k == 1) then
        call fwd%calc_travel_time_single(evt_id, hypo, t_corr, vs, t1)
        call fwd%calc_amp_single(evt_id, hypo, a_corr, qs, vs, a1)
        do j = 1, n_sta
           write(oo, '(2ES26.17E3)') t1(j), a1(j)
        end do
     end if
  end do
end program forward_probe
