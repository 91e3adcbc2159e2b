!> `module cls_model` for the build's own Fortran host code: the parameter-vector type the reference passes
!> around (src/cls_model.f90:5-27 -- nx, prior_type, x, mu, sigma, step_size), with the constructor,
!> setters/getters and prior sampling the step-5 driver uses.  Perturbation and the Metropolis bookkeeping
!> live on the device (htm_step.hpp), so `perturb` is not needed here.
module cls_model
  use htm_random, only: rand_g, rand_r
  implicit none
  private
  public :: model

  type model
     integer :: nx = 0
     integer, allocatable :: prior_type(:)          ! 0 Gaussian, 1 Rayleigh
     double precision, allocatable :: x(:), mu(:), sigma(:), step_size(:)
   contains
     procedure :: set_prior, set_perturb, get_nx, get_x, get_all_x, set_x, generate_model
  end type model

  interface model
     module procedure new_model
  end interface model

contains

  type(model) function new_model(nx, verb) result(m)
    integer, intent(in) :: nx
    logical, intent(in), optional :: verb
    m%nx = nx
    allocate(m%prior_type(nx), source=0)
    allocate(m%x(nx), source=0.d0)
    allocate(m%mu(nx), source=0.d0)
    allocate(m%sigma(nx), source=1.d0)
    allocate(m%step_size(nx), source=0.d0)
  end function new_model

  subroutine set_prior(self, i, mu, sigma, prior_type)
    class(model), intent(inout) :: self
    integer, intent(in) :: i
    double precision, intent(in) :: mu, sigma
    integer, intent(in), optional :: prior_type
    self%mu(i) = mu
    self%sigma(i) = sigma
    self%prior_type(i) = 0
    if (present(prior_type)) self%prior_type(i) = prior_type
  end subroutine set_prior

  subroutine set_perturb(self, i, step_size)
    class(model), intent(inout) :: self
    integer, intent(in) :: i
    double precision, intent(in) :: step_size
    self%step_size(i) = step_size
  end subroutine set_perturb

  pure integer function get_nx(self)
    class(model), intent(in) :: self
    get_nx = self%nx
  end function get_nx

  pure double precision function get_x(self, iparam)
    class(model), intent(in) :: self
    integer, intent(in) :: iparam
    get_x = self%x(iparam)
  end function get_x

  function get_all_x(self) result(all_x)
    class(model), intent(in) :: self
    double precision :: all_x(self%nx)
    all_x = self%x
  end function get_all_x

  subroutine set_x(self, iparam, x)
    class(model), intent(inout) :: self
    integer, intent(in) :: iparam
    double precision, intent(in) :: x
    self%x(iparam) = x
  end subroutine set_x

  !> draw every component from its prior, in index order (the order fixes the random-stream consumption)
  subroutine generate_model(self)
    class(model), intent(inout) :: self
    integer :: i
    do i = 1, self%nx
       select case (self%prior_type(i))
       case (0)
          self%x(i) = self%mu(i) + rand_g() * self%sigma(i)
       case (1)
          self%x(i) = self%mu(i) + rand_r() * self%sigma(i)
       case default
          write(0, '(A,I0,A,I0,A)') "generate_model: element ", i, " has prior_type ", self%prior_type(i), " (known: 0 Gaussian, 1 Rayleigh)"
          stop
       end select
    end do
  end subroutine generate_model

end module cls_model
