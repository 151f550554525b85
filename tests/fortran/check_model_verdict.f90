!=======================================================================
! proposal_is_valid (rf_inv_amd/fortran/model_check.f90) against the verdict of the
! REFERENCE's own format_model (src/model.f90:175-290, compiled unmodified) on random and
! adversarial proposals drawn around the params.in in the working directory:
!   usage: check_model_verdict params.in n_random
! Prints "check_model_verdict: <n> proposals, <valid> valid, <mismatches> mismatches".
! Proposal kinds: the sampler's own (prior draws, one-parameter perturbations of valid models),
! exact ties of two interface depths, thicknesses exactly at / one ulp around h_min and the
! 0.125 * alpha rule, velocities exactly at the limits, interfaces at z_min / z_max.
!=======================================================================
program check_model_verdict
  use params
  use mt19937
  use model
  use math, only: gauss
  use rf_model_check
  implicit none
  character(clen_max) :: param_file, arg
  integer :: n, i, kk, j, kind_, nvalid, nbad, nlay, pick, nmove, other
  logical :: ok_ref, ok_new
  real(8) :: pz(200), pvp(200), pvs(200), alpha(nlay_max), beta(nlay_max), rho(nlay_max), h(nlay_max), u, v
  real(8) :: qz(200), qvp(200), qvs(200)

  param_file = "params.in"
  n = 1000000
  if (command_argument_count() > 0) call get_command_argument(1, param_file)
  if (command_argument_count() > 1) then
     call get_command_argument(2, arg)
     read(arg, *) n
  end if
  call get_params(.false., param_file)
  call sgrnd(iseed)
  call read_ref_model(.false.)
  nvalid = 0
  nbad = 0
  do i = 1, n
     pz = 0.d0; pvp = 0.d0; pvs = 0.d0
     kk = k_min + int(grnd() * (k_max - k_min))
     do j = 1, kk
        pz(j) = z_min + grnd() * (z_max - z_min)
        pvs(j) = gauss() * dvs_prior * 0.3d0
        pvp(j) = gauss() * dvp_prior * 0.3d0
     end do
     pvs(k_max) = gauss() * dvs_prior * 0.3d0
     pvp(k_max) = gauss() * dvp_prior * 0.3d0
     kind_ = mod(i, 8)
     pick = 1 + int(grnd() * kk)
     select case (kind_)
     case (1)                       ! two interfaces at exactly the same depth
        if (kk >= 2) pz(pick) = pz(1 + mod(pick, kk))
     case (2)                       ! a middle layer exactly h_min thick, or one ulp thinner / thicker
        if (kk >= 2) then
           u = pz(1 + mod(pick, kk)) + h_min
           if (mod(i / 8, 3) == 1) u = nearest(u, 1.d0)
           if (mod(i / 8, 3) == 2) u = nearest(u, -1.d0)
           pz(pick) = u
        end if
     case (3)                       ! the shallowest interface around the 0.125 * alpha rule
        u = sdep + 0.125d0 * vp_ref(1)
        if (mod(i / 8, 3) == 1) u = nearest(u, 1.d0)
        if (mod(i / 8, 3) == 2) u = nearest(u, -1.d0)
        pz(pick) = u
     case (4)                       ! a velocity exactly at a limit
        if (mod(i / 8, 2) == 0) then
           pvs(pick) = vs_max - vs_ref(1)
        else
           pvs(pick) = vs_min - vs_ref(1)
        end if
     case (5)                       ! interfaces at the ends of the depth range
        pz(pick) = merge(z_min, z_max, mod(i / 8, 2) == 0)
     case (6)                       ! large perturbations: the ratio and range rules
        pvs(pick) = gauss() * 2.d0
        pvp(pick) = gauss() * 1.d0
     case default                   ! the sampler's own kind of proposal
     end select
     call format_model(kk, pz(1:k_max-1), pvp(1:k_max), pvs(1:k_max), nlay, alpha, beta, rho, h, ok_ref)
     ok_new = proposal_is_valid(kk, pz(1:k_max-1), pvp(1:k_max), pvs(1:k_max))
     if (ok_ref) nvalid = nvalid + 1
     if (ok_ref .neqv. ok_new) then
        nbad = nbad + 1
        if (nbad <= 5) write(*,*) "MISMATCH kind", kind_, " k", kk, " reference", ok_ref, " ours", ok_new
     end if
  end do
  write(*,'(A,I0,A,I0,A,I0,A)') " check_model_verdict: ", n, " proposals, ", nvalid, " valid, ", nbad, " mismatches"

  ! ---- velocity_move_is_valid: a VALID model + a change of one velocity perturbation (what a dVs / dVp move of the
  ! sampler is), against the reference's format_model on the changed model.  Moves: the sampler's small steps, steps
  ! that land exactly on / one ulp beyond a velocity limit, large ones (ratio rules), the half-space slot, the top
  ! layer (whose thickness rule depends on its velocity when vp_mode = 1).
  nvalid = 0
  nbad = 0
  nmove = 0
  do i = 1, n
     pz = 0.d0; pvp = 0.d0; pvs = 0.d0
     kk = k_min + int(grnd() * (k_max - k_min))
     do j = 1, kk
        pz(j) = z_min + grnd() * (z_max - z_min)
        pvs(j) = gauss() * dvs_prior * 0.2d0
        pvp(j) = gauss() * dvp_prior * 0.2d0
     end do
     pvs(k_max) = gauss() * dvs_prior * 0.2d0
     pvp(k_max) = gauss() * dvp_prior * 0.2d0
     call format_model(kk, pz(1:k_max-1), pvp(1:k_max), pvs(1:k_max), nlay, alpha, beta, rho, h, ok_ref)
     if (.not. ok_ref) cycle                      ! the chain's current model is always valid
     do j = 1, 4
        pick = 1 + int(grnd() * (kk + 1))
        if (pick == kk + 1) pick = k_max
        u = pvs(pick)
        v = pvp(pick)
        select case (mod(i + j, 6))
        case (0)
           pvs(pick) = pvs(pick) + gauss() * dev_dvs
        case (1)
           pvp(pick) = pvp(pick) + gauss() * dev_dvp
        case (2)
           pvs(pick) = pvs(pick) + gauss() * 1.5d0
        case (3)
           pvp(pick) = pvp(pick) + gauss() * 1.0d0
        case (4)                                  ! exactly at / one ulp beyond a limit of Vs
           pvs(pick) = merge(vs_max, vs_min, mod(i, 2) == 0) - vs_ref(1)
           if (mod(i / 2, 3) == 1) pvs(pick) = nearest(pvs(pick), 1.d0)
           if (mod(i / 2, 3) == 2) pvs(pick) = nearest(pvs(pick), -1.d0)
        case default                              ! the Vp / Vs ratio at its limits
           pvs(pick) = vp_ref(1) / merge(vpvs_max, vpvs_min, mod(i, 2) == 0) - vs_ref(1)
           if (mod(i / 2, 3) == 1) pvs(pick) = nearest(pvs(pick), 1.d0)
           if (mod(i / 2, 3) == 2) pvs(pick) = nearest(pvs(pick), -1.d0)
        end select
        call format_model(kk, pz(1:k_max-1), pvp(1:k_max), pvs(1:k_max), nlay, alpha, beta, rho, h, ok_ref)
        ok_new = velocity_move_is_valid(kk, pz(1:k_max-1), pvp(1:k_max), pvs(1:k_max), pick)
        nmove = nmove + 1
        if (ok_ref) nvalid = nvalid + 1
        if (ok_ref .neqv. ok_new) then
           nbad = nbad + 1
           if (nbad <= 5) write(*,*) "MOVE MISMATCH k", kk, " pick", pick, " reference", ok_ref, " ours", ok_new
        end if
        pvs(pick) = u                             ! back to the valid model
        pvp(pick) = v
     end do
  end do
  write(*,'(A,I0,A,I0,A,I0,A)') " check_model_verdict: ", nmove, " velocity moves, ", nvalid, " valid, ", nbad, " mismatches"

  ! ---- interface_move_is_valid / interface_removal_is_valid: a VALID model + a depth move, a birth or a death, against
  ! the reference's format_model on the changed model.  Depth moves: the sampler's small steps, large ones (the
  ! interface passes others), landings exactly at / one ulp around h_min from a neighbour and the 0.125 alpha rule,
  ! exact ties.  Births: anywhere, at h_min (+- one ulp) from an existing interface, on top of one.  Deaths: any.
  nvalid = 0
  nbad = 0
  nmove = 0
  do i = 1, n
     pz = 0.d0; pvp = 0.d0; pvs = 0.d0
     kk = max(k_min, 1) + int(grnd() * (k_max - 1 - max(k_min, 1)))       ! room for a birth: kk + 1 < k_max
     do j = 1, kk
        pz(j) = z_min + grnd() * (z_max - z_min)
        pvs(j) = gauss() * dvs_prior * 0.2d0
        pvp(j) = gauss() * dvp_prior * 0.2d0
     end do
     pvs(k_max) = gauss() * dvs_prior * 0.2d0
     pvp(k_max) = gauss() * dvp_prior * 0.2d0
     call format_model(kk, pz(1:k_max-1), pvp(1:k_max), pvs(1:k_max), nlay, alpha, beta, rho, h, ok_ref)
     if (.not. ok_ref) cycle
     do j = 1, 6
        qz = pz; qvp = pvp; qvs = pvs
        kind_ = mod(i + j, 12)
        pick = 1 + int(grnd() * kk)
        other = 1 + mod(pick + int(grnd() * max(kk - 1, 1)), kk)
        if (kind_ <= 5) then
           ! a depth move of interface pick
           select case (kind_)
           case (0, 1)
              qz(pick) = pz(pick) + gauss() * dev_z
           case (2)
              qz(pick) = pz(pick) + gauss() * 8.d0
           case (3)
              u = pz(other) + merge(h_min, -h_min, mod(i, 2) == 0)
              if (mod(i / 2, 3) == 1) u = nearest(u, 1.d0)
              if (mod(i / 2, 3) == 2) u = nearest(u, -1.d0)
              qz(pick) = u
           case (4)
              u = sdep + 0.125d0 * vp_ref(1)
              if (mod(i / 2, 3) == 1) u = nearest(u, 1.d0)
              if (mod(i / 2, 3) == 2) u = nearest(u, -1.d0)
              qz(pick) = u
           case default
              if (kk >= 2) qz(pick) = pz(other)
           end select
           if (qz(pick) < z_min .or. qz(pick) > z_max) cycle          ! (the sampler drops such a proposal before the verdict)
           call format_model(kk, qz(1:k_max-1), qvp(1:k_max), qvs(1:k_max), nlay, alpha, beta, rho, h, ok_ref)
           ok_new = interface_move_is_valid(kk, qz(1:k_max-1), qvp(1:k_max), qvs(1:k_max), pick, pz(pick), .true.)
        else if (kind_ <= 9) then
           ! a birth: interface kk + 1
           select case (kind_)
           case (6, 7)
              u = z_min + grnd() * (z_max - z_min)
           case (8)
              u = pz(pick) + merge(h_min, -h_min, mod(i, 2) == 0)
              if (mod(i / 2, 3) == 1) u = nearest(u, 1.d0)
              if (mod(i / 2, 3) == 2) u = nearest(u, -1.d0)
           case default
              u = pz(pick)
           end select
           if (u < z_min .or. u > z_max) cycle
           qz(kk + 1) = u
           qvs(kk + 1) = gauss() * dvs_prior * merge(0.2d0, 1.d0, mod(i, 3) /= 0)
           qvp(kk + 1) = gauss() * dvp_prior * merge(0.2d0, 1.d0, mod(i, 3) /= 0)
           call format_model(kk + 1, qz(1:k_max-1), qvp(1:k_max), qvs(1:k_max), nlay, alpha, beta, rho, h, ok_ref)
           ok_new = interface_move_is_valid(kk + 1, qz(1:k_max-1), qvp(1:k_max), qvs(1:k_max), kk + 1, 0.d0, .false.)
        else
           ! a death of interface pick (the sampler's shift, src/pt_mcmc.f90:108-124)
           if (kk - 1 < max(k_min, 1)) cycle
           if (pick < kk) then
              qz(pick:kk-1) = pz(pick+1:kk)
              qvp(pick:kk-1) = pvp(pick+1:kk)
              qvs(pick:kk-1) = pvs(pick+1:kk)
           end if
           qz(kk) = 0.d0; qvp(kk) = 0.d0; qvs(kk) = 0.d0
           call format_model(kk - 1, qz(1:k_max-1), qvp(1:k_max), qvs(1:k_max), nlay, alpha, beta, rho, h, ok_ref)
           ok_new = interface_removal_is_valid(kk - 1, qz(1:k_max-1), qvp(1:k_max), qvs(1:k_max), pz(pick))
        end if
        nmove = nmove + 1
        if (ok_ref) nvalid = nvalid + 1
        if (ok_ref .neqv. ok_new) then
           nbad = nbad + 1
           if (nbad <= 8) write(*,*) "INTERFACE MISMATCH kind", kind_, " k", kk, " pick", pick, " reference", ok_ref, " ours", ok_new
        end if
     end do
  end do
  write(*,'(A,I0,A,I0,A,I0,A)') " check_model_verdict: ", nmove, " interface moves, ", nvalid, " valid, ", nbad, " mismatches"
end program check_model_verdict
