!=======================================================================
! module rf_model_check -- the VERDICT of the reference's format_model
! (src/model.f90:175-290: is the proposed model valid?) without its products.
!
! pt_control_batched needs that verdict on the host -- a chain's random
! stream depends on it (an invalid model is a null proposal: no acceptance
! uniform is drawn, src/pt_mcmc.f90:163-176) -- but nothing else of
! format_model: the engine formats the layer stack itself, bit for bit
! (format_model_kernel).  The reference routine costs ~0.35-0.55 us per call
! at k_max 30 (a recursive three-array quicksort, the density polynomial of
! every layer, five output arrays), which made the host's proposal loop the
! bound of the whole sampler; this one sorts an index by insertion and stops
! at the first violated rule.
!
! Same verdict as format_model for every input the sampler can produce:
!   * the layer rules are the reference's expressions, in its operand order
!     (:219-231 top layer incl. the `0.125 * alpha` thickness rule, :248-260
!     middle layers, :276-281 half-space; nint look-ups :212,239,268);
!   * the sort only has to ORDER the interfaces: two equal depths would make
!     the reference's unstable quicksort decide which perturbation belongs to
!     which layer, but they also make a layer of thickness 0 < h_min, i.e. an
!     invalid model whatever the order.  With h_min <= 0 that argument fails
!     and the reference routine is called instead.
! Checked against format_model itself on millions of random and adversarial
! proposals (tests/fortran/check_model_verdict.f90, tests/test_host_model.py).
!=======================================================================
module rf_model_check
  implicit none
  public proposal_is_valid, velocity_move_is_valid, interface_move_is_valid, interface_removal_is_valid
  private

contains

  logical function proposal_is_valid(prop_k, prop_z, prop_dvp, prop_dvs) result(ok)
    use params, only: k_max, nlay_max, sdep, z_max, h_min, vp_mode, vp_min, vp_max, vs_min, vs_max, vpvs_min, vpvs_max
    use model, only: format_model, vp_ref, vs_ref, z_ref_min, dz_ref
    integer, intent(in) :: prop_k
    real(8), intent(in) :: prop_z(k_max-1), prop_dvp(k_max), prop_dvs(k_max)
    integer :: idx(k_max), i, j, t, iz, nlay
    real(8) :: zj, zprev, zc, a, b, thick
    real(8) :: alpha(nlay_max), beta(nlay_max), rho(nlay_max), h(nlay_max)

    if (.not. (h_min > 0.d0)) then
       call format_model(prop_k, prop_z, prop_dvp, prop_dvs, nlay, alpha, beta, rho, h, ok)
       return
    end if

    ! interfaces in ascending depth (insertion sort of an index: k <= 29, mostly short)
    do i = 1, prop_k
       t = i
       zj = prop_z(i)
       j = i - 1
       do while (j >= 1)
          if (prop_z(idx(j)) <= zj) exit
          idx(j + 1) = idx(j)
          j = j - 1
       end do
       idx(j + 1) = t
    end do

    ok = .false.
    zprev = sdep
    do j = 1, prop_k + 1
       if (j <= prop_k) then
          zj = prop_z(idx(j))
          zc = 0.5d0 * (zj + zprev)            ! :210 (top: 0.5 (sdep + z1)), :238
          t = idx(j)
       else
          zj = z_max
          zc = 0.5d0 * (z_max + zprev)         ! :267
          t = k_max
       end if
       iz = nint((zc - z_ref_min) / dz_ref) + 1
       b = vs_ref(iz) + prop_dvs(t)
       if (vp_mode == 1) then
          a = vp_ref(iz) + prop_dvp(t)
       else
          a = vp_ref(iz)
       end if
       if (a < vp_min .or. a > vp_max .or. b < vs_min .or. b > vs_max .or. &
            & a / b < vpvs_min .or. a / b > vpvs_max) return
       if (j == 1) then
          thick = zj - sdep
          if (thick < 0.125 * a) return        ! :229 (not h_min)
       else if (j <= prop_k) then
          thick = zj - zprev
          if (thick < h_min) return            ! :256
       end if
       zprev = zj
    end do
    ok = .true.
  end function proposal_is_valid

  !---------------------------------------------------------------------
  ! The verdict for a proposal that changes ONE velocity perturbation (dVs or dVp of slot `pick`: an interface
  ! 1 .. prop_k, or k_max = the half-space) of a model that is VALID without the change -- a chain's current model
  ! always is.  format_model's rules are per layer (src/model.f90:219-231, :248-260, :276-281) and every other
  ! layer keeps its depth, thickness and velocities, so only the layer that owns the slot is examined: its place in
  ! depth order and the interface above it come from one pass over the depths -- no sort, no other layer's
  ! look-ups.  Same expressions, in the same operand order, as proposal_is_valid for that layer.  (With h_min <= 0,
  ! where two interfaces may coincide and the reference's unstable sort decides which perturbation belongs to which
  ! layer, the full check is used.)
  logical function velocity_move_is_valid(prop_k, prop_z, prop_dvp, prop_dvs, pick) result(ok)
    use params, only: k_max, sdep, z_max, h_min, vp_mode, vp_min, vp_max, vs_min, vs_max, vpvs_min, vpvs_max
    use model, only: vp_ref, vs_ref, z_ref_min, dz_ref
    integer, intent(in) :: prop_k, pick
    real(8), intent(in) :: prop_z(k_max-1), prop_dvp(k_max), prop_dvs(k_max)
    integer :: i, iz
    logical :: top
    real(8) :: zj, zprev, zc, a, b

    if (.not. (h_min > 0.d0)) then
       ok = proposal_is_valid(prop_k, prop_z, prop_dvp, prop_dvs)
       return
    end if
    ok = .false.
    zprev = sdep
    top = .true.
    if (pick <= prop_k) then
       ! the interface just above this one (the deepest of the shallower ones), or the surface / sea floor
       zj = prop_z(pick)
       do i = 1, prop_k
          if (prop_z(i) < zj) then
             if (top) then
                zprev = prop_z(i)
                top = .false.
             else
                zprev = max(zprev, prop_z(i))
             end if
          end if
       end do
       zc = 0.5d0 * (zj + zprev)
    else
       ! the half-space: below the deepest interface
       do i = 1, prop_k
          if (top) then
             zprev = prop_z(i)
             top = .false.
          else
             zprev = max(zprev, prop_z(i))
          end if
       end do
       zj = z_max
       zc = 0.5d0 * (z_max + zprev)
       top = .false.
    end if
    iz = nint((zc - z_ref_min) / dz_ref) + 1
    b = vs_ref(iz) + prop_dvs(pick)
    if (vp_mode == 1) then
       a = vp_ref(iz) + prop_dvp(pick)
    else
       a = vp_ref(iz)
    end if
    if (a < vp_min .or. a > vp_max .or. b < vs_min .or. b > vs_max .or. &
         & a / b < vpvs_min .or. a / b > vpvs_max) return
    if (top) then
       if (zj - sdep < 0.125 * a) return        ! :229: the top layer's thickness rule depends on its velocity
    end if
    ok = .true.
  end function velocity_move_is_valid

  !---------------------------------------------------------------------
  ! The rules of ONE layer -- the one whose bottom is the interface at depth zj (or z_max: the half-space) with the
  ! interface zprev above it (sdep for the top layer) and the perturbations of slot islot -- exactly as
  ! proposal_is_valid evaluates them (src/model.f90:210-231 top, :238-260 middle, :267-281 half-space).
  logical function layer_is_valid(zj, zprev, islot, top, halfspace, prop_dvp, prop_dvs) result(ok)
    use params, only: k_max, sdep, h_min, vp_mode, vp_min, vp_max, vs_min, vs_max, vpvs_min, vpvs_max
    use model, only: vp_ref, vs_ref, z_ref_min, dz_ref
    real(8), intent(in) :: zj, zprev, prop_dvp(k_max), prop_dvs(k_max)
    integer, intent(in) :: islot
    logical, intent(in) :: top, halfspace
    integer :: iz
    real(8) :: zc, a, b, thick

    ok = .false.
    zc = 0.5d0 * (zj + zprev)
    iz = nint((zc - z_ref_min) / dz_ref) + 1
    b = vs_ref(iz) + prop_dvs(islot)
    if (vp_mode == 1) then
       a = vp_ref(iz) + prop_dvp(islot)
    else
       a = vp_ref(iz)
    end if
    if (a < vp_min .or. a > vp_max .or. b < vs_min .or. b > vs_max .or. &
         & a / b < vpvs_min .or. a / b > vpvs_max) return
    if (top) then
       thick = zj - sdep
       if (thick < 0.125 * a) return
    else if (.not. halfspace) then
       thick = zj - zprev
       if (thick < h_min) return
    end if
    ok = .true.
  end function layer_is_valid

  !---------------------------------------------------------------------
  ! The verdict for a proposal that MOVES interface `pick` of a valid model from z_old to prop_z(pick) (moved), or
  ! ADDS interface `pick` = prop_k at prop_z(pick) (.not. moved: a birth).  Without the interface the model is valid and
  ! every layer that does not touch the new depth keeps its depth range, owner and velocities; two layers do: the one
  ! above the new depth (owned by the interface itself) and the one below it (owned by the next interface down, or the
  ! half-space), whose centre and thickness change.  A move that passes another interface changes which perturbation
  ! belongs to which layer further away: such a proposal (and h_min <= 0) takes the full check.
  logical function interface_move_is_valid(prop_k, prop_z, prop_dvp, prop_dvs, pick, z_old, moved) result(ok)
    use params, only: k_max, sdep, z_max, h_min
    integer, intent(in) :: prop_k, pick
    real(8), intent(in) :: prop_z(k_max-1), prop_dvp(k_max), prop_dvs(k_max), z_old
    logical, intent(in) :: moved
    integer :: i, inext, nbefore_new, nbefore_old
    real(8) :: znew, zi, zprev, znext

    if (.not. (h_min > 0.d0)) then
       ok = proposal_is_valid(prop_k, prop_z, prop_dvp, prop_dvs)
       return
    end if
    ok = .false.
    znew = prop_z(pick)
    zprev = sdep
    inext = 0
    znext = z_max
    nbefore_new = 0
    nbefore_old = 0
    do i = 1, prop_k
       if (i == pick) cycle
       zi = prop_z(i)
       if (zi < znew) then
          if (nbefore_new == 0) then
             zprev = zi
          else
             zprev = max(zprev, zi)
          end if
          nbefore_new = nbefore_new + 1
       else if (zi > znew) then
          if (inext == 0) then
             inext = i
             znext = zi
          else if (zi < znext) then
             inext = i
             znext = zi
          end if
       else
          return                        ! two interfaces at one depth: a layer of thickness 0 < h_min
       end if
       if (moved .and. zi < z_old) nbefore_old = nbefore_old + 1
    end do
    if (moved .and. nbefore_old /= nbefore_new) then
       ok = proposal_is_valid(prop_k, prop_z, prop_dvp, prop_dvs)
       return
    end if
    ! the layer above the new depth, then the one below it (in the reference's order: the shallower layer first)
    if (.not. layer_is_valid(znew, zprev, pick, nbefore_new == 0, .false., prop_dvp, prop_dvs)) return
    if (inext > 0) then
       ok = layer_is_valid(znext, znew, inext, .false., .false., prop_dvp, prop_dvs)
    else
       ok = layer_is_valid(z_max, znew, k_max, .false., .true., prop_dvp, prop_dvs)
    end if
  end function interface_move_is_valid

  !---------------------------------------------------------------------
  ! The verdict for a proposal that REMOVES the interface that was at z_removed from a valid model (a death: the
  ! arrays passed are the ones after the removal).  The layer it bounded merges with the one below, which now reaches
  ! up to the interface above: only that layer -- owned by the next interface down, or the half-space -- changes.
  logical function interface_removal_is_valid(prop_k, prop_z, prop_dvp, prop_dvs, z_removed) result(ok)
    use params, only: k_max, sdep, z_max, h_min
    integer, intent(in) :: prop_k
    real(8), intent(in) :: prop_z(k_max-1), prop_dvp(k_max), prop_dvs(k_max), z_removed
    integer :: i, inext, nbefore
    real(8) :: zi, zprev, znext

    if (.not. (h_min > 0.d0)) then
       ok = proposal_is_valid(prop_k, prop_z, prop_dvp, prop_dvs)
       return
    end if
    zprev = sdep
    inext = 0
    znext = z_max
    nbefore = 0
    do i = 1, prop_k
       zi = prop_z(i)
       if (zi < z_removed) then
          if (nbefore == 0) then
             zprev = zi
          else
             zprev = max(zprev, zi)
          end if
          nbefore = nbefore + 1
       else
          if (inext == 0) then
             inext = i
             znext = zi
          else if (zi < znext) then
             inext = i
             znext = zi
          end if
       end if
    end do
    if (inext > 0) then
       ok = layer_is_valid(znext, zprev, inext, nbefore == 0, .false., prop_dvp, prop_dvs)
    else
       ok = layer_is_valid(z_max, zprev, k_max, .false., .true., prop_dvp, prop_dvs)
    end if
  end function interface_removal_is_valid

end module rf_model_check
