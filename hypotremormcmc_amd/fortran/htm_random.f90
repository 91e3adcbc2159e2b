!> Host-side copy of the rank's random stream generator, used for the set-up draws only (initial models
!> and temperatures); the iteration draws are produced on the device from the state handed over at chain
!> creation.  Algorithm = the reference's mod_random (src/mod_random.f90): Marsaglia xorshift128 on four
!> 32-bit words, uniform deviates from the last word, Box-Muller (cosine branch), Rayleigh by inversion.
module htm_random
  use, intrinsic :: iso_fortran_env, only: int32, int64, real64
  implicit none
  private
  public :: rng_seed, rng_state, rand_u, rand_u2, rand_g, rand_r

  integer(int32), save :: s(4) = 0      ! (x, y, z, w)
  real(real64), parameter :: two31 = 2147483648.0_real64, two32 = 4294967296.0_real64

contains

  !> seeds mixed with the rank exactly like init_random (wrapping 32-bit arithmetic)
  subroutine rng_seed(seeds, rank)
    integer, intent(in) :: seeds(4), rank
    integer(int64) :: j, j2, j4, v
    integer :: k
    j = int(rank, int64) + 1_int64
    j2 = wrap(j * j)
    j4 = wrap(j2 * j2)
    do k = 1, 4
       v = wrap(wrap(int(seeds(k), int64) * j4) + wrap(wrap(1000_int64 * int(seeds(k), int64)) * j2) &
            & + int(seeds(k), int64))
       s(k) = int(v, int32)
    end do
  contains
    !> two's-complement wrap of a 64-bit value to the signed 32-bit range
    pure function wrap(a) result(r)
      integer(int64), intent(in) :: a
      integer(int64) :: r
      r = iand(a, 4294967295_int64)
      if (r >= 2147483648_int64) r = r - 4294967296_int64
    end function wrap
  end subroutine rng_seed

  function rng_state() result(st)
    integer(int32) :: st(4)
    st = s
  end function rng_state

  !> advance the generator, return the new last word
  function next_word() result(w)
    integer(int32) :: w, t
    t = ieor(s(1), shiftl(s(1), 11))
    s(1:3) = s(2:4)
    s(4) = ieor(ieor(s(4), shiftr(s(4), 19)), ieor(t, shiftr(t, 8)))
    w = s(4)
  end function next_word

  function rand_u() result(u)      ! [0, 1)
    real(real64) :: u
    u = (real(next_word(), real64) + two31) / two32
  end function rand_u

  function rand_u2() result(u)     ! (0, 1)
    real(real64) :: u
    u = (real(next_word(), real64) + two31 + 0.5_real64) / two32
  end function rand_u2

  function rand_g() result(g)
    real(real64) :: g, a, b
    real(real64), parameter :: two_pi = 2.0_real64 * acos(-1.0_real64)
    a = rand_u2()
    b = rand_u2()
    g = sqrt(-2.0_real64 * log(a)) * cos(two_pi * b)
  end function rand_g

  function rand_r() result(r)
    real(real64) :: r
    r = sqrt(-2.0_real64 * log(rand_u2()))
  end function rand_r

end module htm_random
