# tools/dump_goldens.jl -- turns a `CPU()` run of the reference (Oceananigans =0.96.26 through the GordonBell25 API)
# into the golden vectors that would PIN this repository's oracle (DESIGN.md section 0: "parity unpinned").
#
#   julia --project=<GB-25 checkout> tools/dump_goldens.jl [output directory, default tests/golden/julia]
#
# For each case (the reference's own correctness configuration 112x112x16 with dt = 1e-9, and BASELINE config 1
# 128x64x8 with dt = 1200 s from the deterministic baroclinic state) and each float type (Float32, Float64) it follows
# correctness/correctness_baroclinic_instability_simulation_run.jl:40-102 on the CPU model alone and writes, at each of
# its six checkpoints, `parent(field)` of every field compare_states walks (src/correctness.jl:28-90): u, v, w, eta, T,
# S, G^n and G^- of u, v, T, S, and the split-explicit filtered state, plus the barotropic U, V and the metrics.
# Output: <dir>/<case>_<FT>/<checkpoint>/<field>.npy (NumPy .npy v1.0, fortran_order, so np.load gives [i, j, k]).
# tests/test_julia_goldens.py consumes them (oracle AND HIP) and skips while the directory is empty.
#
# Velocity noise comes from the build-owned counter RNG (seed 42), not from Julia's rand: the stream of
# Random.seed!(42) cannot be reproduced outside Julia, and the first checkpoint stores the state anyway.
#
# NOT EXECUTED in the build image (no Julia there).
using GordonBell25
using Oceananigans
using Printf

function write_npy(path::AbstractString, A::AbstractArray{T}) where {T<:Union{Float32,Float64}}
    descr = T === Float32 ? "<f4" : "<f8"
    shape = join(string.(size(A)), ", ") * (ndims(A) == 1 ? "," : "")
    header = "{'descr': '$descr', 'fortran_order': True, 'shape': ($shape), }"
    pad = 64 - mod(10 + length(header) + 1, 64)          # magic(6) + version(2) + length(2) + header + '\n'
    header = header * " "^(pad == 64 ? 0 : pad) * "\n"
    open(path, "w") do io
        write(io, UInt8[0x93, UInt8('N'), UInt8('U'), UInt8('M'), UInt8('P'), UInt8('Y'), 0x01, 0x00])
        write(io, htol(UInt16(length(header))))
        write(io, header)
        write(io, Array(A))                              # column-major bytes = fortran_order
    end
end

function counter_rng(dims::NTuple{N,Int}, seed::Integer, salt::Integer) where {N}
    out = Array{Float64}(undef, dims)
    @inbounds for q in 0:prod(dims)-1
        x = UInt64(q) + UInt64(seed) * 0x9E3779B97F4A7C15 + UInt64(salt) * 0xD1B54A32D192ED03
        x ⊻= x >> 30; x *= 0xBF58476D1CE4E5B9
        x ⊻= x >> 27; x *= 0x94D049BB133111EB
        x ⊻= x >> 31
        out[q + 1] = Float64(x >> 11) / Float64(UInt64(1) << 53)
    end
    return out
end

function dump_checkpoint(dir, model)
    mkpath(dir)
    Ψ = Oceananigans.fields(model)
    for name in keys(Ψ)
        write_npy(joinpath(dir, "$(name).npy"), parent(Ψ[name]))
        if !(name ∈ (:w, :η))
            write_npy(joinpath(dir, "Gn.$(name).npy"), parent(model.timestepper.Gⁿ[name]))
            write_npy(joinpath(dir, "Gm.$(name).npy"), parent(model.timestepper.G⁻[name]))
        end
    end
    fs = model.free_surface
    for name in (:U, :V, :η)
        write_npy(joinpath(dir, "filtered.$(name).npy"), parent(getproperty(fs.filtered_state, name)))
    end
    write_npy(joinpath(dir, "U.npy"), parent(fs.barotropic_velocities.U))
    write_npy(joinpath(dir, "V.npy"), parent(fs.barotropic_velocities.V))
    write_npy(joinpath(dir, "pHY.npy"), parent(model.pressure.pHY′))
    # closure = CATKEVerticalDiffusivity(): the fields src/correctness.jl:60-67 compares (e, Gn.e, Gm.e are in fields(model))
    κ = model.diffusivity_fields
    if κ isa NamedTuple && haskey(κ, :κu)
        for name in (:κu, :κc, :κe, :Le, :Jᵇ)
            write_npy(joinpath(dir, "$(name).npy"), parent(getproperty(κ, name)))
        end
    end
    open(joinpath(dir, "clock.txt"), "w") do io
        @printf(io, "time %.17g\niteration %d\nlast_dt %.17g\n", model.clock.time, model.clock.iteration, model.clock.last_Δt)
    end
end

function dump_grid(dir, grid)
    mkpath(dir)
    FT = eltype(grid)
    for (name, a) in (("zf", grid.z.cᵃᵃᶠ), ("zc", grid.z.cᵃᵃᶜ), ("dzc", grid.z.Δᵃᵃᶜ), ("dzf", grid.z.Δᵃᵃᶠ),
                      ("phif", grid.φᵃᶠᵃ), ("phic", grid.φᵃᶜᵃ), ("dxc", grid.Δxᶠᶜᵃ), ("dxf", grid.Δxᶜᶠᵃ),
                      ("azc", grid.Azᶜᶜᵃ), ("azf", grid.Azᶠᶠᵃ))
        write_npy(joinpath(dir, "$(name).npy"), FT.(collect(parent(a))))
    end
end

# the 2-D metrics of an orthogonal curvilinear underlying grid (TripolarGrid), the coordinates of the cell centres and
# the bottom height: what pins this repository's analytic tripolar restatement (DESIGN.md section 0) -- or shows how far
# its cap is from Oceananigans' numerically generated one
function dump_curvilinear_grid(dir, ibg)
    mkpath(dir)
    grid = ibg.underlying_grid
    FT = eltype(grid)
    for name in (:Δxᶠᶜᵃ, :Δxᶜᶜᵃ, :Δxᶜᶠᵃ, :Δxᶠᶠᵃ, :Δyᶠᶜᵃ, :Δyᶜᶜᵃ, :Δyᶜᶠᵃ, :Δyᶠᶠᵃ, :Azᶜᶜᵃ, :Azᶠᶜᵃ, :Azᶜᶠᵃ, :Azᶠᶠᵃ,
                 :λᶜᶜᵃ, :φᶜᶜᵃ, :λᶠᶠᵃ, :φᶠᶠᵃ)
        write_npy(joinpath(dir, "$(name).npy"), FT.(collect(parent(getproperty(grid, name)))))
    end
    for (name, a) in (("zf", grid.z.cᵃᵃᶠ), ("zc", grid.z.cᵃᵃᶜ), ("dzc", grid.z.Δᵃᵃᶜ), ("dzf", grid.z.Δᵃᵃᶠ))
        write_npy(joinpath(dir, "$(name).npy"), FT.(collect(parent(a))))
    end
    write_npy(joinpath(dir, "bottom_height.npy"), FT.(collect(parent(ibg.immersed_boundary.bottom_height))))
end

# kw: what the case passes on to baroclinic_instability_model (grid_type = :gaussian_islands;
# closure = VerticalScalarDiffusivity(VerticallyImplicitTimeDiscretization(), κ = ..., ν = ...): both are keyword arguments
# of the reference's own constructor, src/baroclinic_instability_model.jl:17-31)
function run_case(outdir, casename, FT, Nx, Ny, Nz, Δt, baroclinic_state::Bool; kw...)
    Oceananigans.defaults.FloatType = FT
    model = GordonBell25.baroclinic_instability_model(CPU(), Nx, Ny, Nz; Δt, halo = (8, 8, 8), kw...)
    dir = joinpath(outdir, "$(casename)_$(FT)")
    if model.grid isa Oceananigans.ImmersedBoundaries.ImmersedBoundaryGrid
        dump_curvilinear_grid(joinpath(dir, "grid"), model.grid)
    else
        dump_grid(joinpath(dir, "grid"), model.grid)
    end
    # split-explicit substepping as materialised by the model (weights, effective substep count, fractional step)
    ss = model.free_surface.substepping
    write_npy(joinpath(dir, "grid", "substep_weights.npy"), Float64.(collect(ss.averaging_weights)))
    open(joinpath(dir, "grid", "substepping.txt"), "w") do io
        @printf(io, "fractional_step_size %.17g\nn_weights %d\n", ss.fractional_step_size, length(ss.averaging_weights))
    end

    baroclinic_state && GordonBell25.set_baroclinic_instability!(model)
    if haskey(Oceananigans.fields(model), :e)            # a CATKE case: some TKE to start from, seeded like u, v
        set!(model, e = 1e-4 .* counter_rng(size(model.tracers.e), 42, 3))
    end
    ui = 1e-3 .* counter_rng(size(model.velocities.u), 42, 1)
    vi = 1e-3 .* counter_rng(size(model.velocities.v), 42, 2)
    set!(model, u = ui, v = vi)
    dump_checkpoint(joinpath(dir, "1_beginning"), model)

    Oceananigans.initialize!(model)
    Oceananigans.TimeSteppers.update_state!(model)
    dump_checkpoint(joinpath(dir, "2_after_initialize_and_update_state"), model)

    GordonBell25.first_time_step!(model)
    dump_checkpoint(joinpath(dir, "3_after_first_time_step"), model)

    for _ in 1:12                                        # 2 warm-up + 10 steps
        GordonBell25.time_step!(model)
    end
    dump_checkpoint(joinpath(dir, "4_after_2_plus_10_steps"), model)

    Oceananigans.TimeSteppers.update_state!(model)       # (the sync of the reference's two models is the identity here)
    dump_checkpoint(joinpath(dir, "5_after_sync_and_update_state"), model)

    GordonBell25.loop!(model, 100)
    dump_checkpoint(joinpath(dir, "6_after_loop_100"), model)
    @info "wrote $dir"
end

# The data-free climate model (src/data_free_ocean_climate_model.jl:12-70): the ocean's state and the four top flux
# boundary conditions the coupled model fills (ocean_simulation gives u, v, T, S a FluxBoundaryCondition over a Field), at
# iteration 0 (after update_state! of the coupled model) and after a few coupled steps.  What pins -- or refutes -- the
# similarity-theory restatement of this repository (DESIGN.md section 4, "Data-free forcing").
function dump_coupled(dir, coupled)
    ocean = coupled.ocean.model
    dump_checkpoint(dir, ocean)
    for (name, f) in (("Ju", ocean.velocities.u), ("Jv", ocean.velocities.v), ("JT", ocean.tracers.T), ("JS", ocean.tracers.S))
        write_npy(joinpath(dir, "$(name).npy"), parent(f.boundary_conditions.top.condition))
    end
end
function run_data_free(outdir, FT, resolution, Nz)
    Oceananigans.defaults.FloatType = FT
    coupled = GordonBell25.data_free_ocean_climate_model_init(CPU(); resolution, Nz)
    dir = joinpath(outdir, "datafree_r$(resolution)x$(Nz)_$(FT)")
    dump_curvilinear_grid(joinpath(dir, "grid"), coupled.ocean.model.grid)
    Oceananigans.TimeSteppers.update_state!(coupled)
    dump_coupled(joinpath(dir, "1_beginning"), coupled)
    for _ in 1:5
        Oceananigans.TimeSteppers.time_step!(coupled, 30)
    end
    dump_coupled(joinpath(dir, "2_after_5_coupled_steps"), coupled)
    @info "wrote $dir"
end

function main(args)
    outdir = length(args) >= 1 ? args[1] : joinpath(@__DIR__, "..", "tests", "golden", "julia")
    for FT in (Float32, Float64)
        run_case(outdir, "protocol_112x112x16", FT, 112, 112, 16, 1e-9, false)
        run_case(outdir, "config1_128x64x8", FT, 128, 64, 8, 1200.0, true)
        # grid_type = :gaussian_islands: TripolarGrid + GridFittedBottom (src/model_utils.jl:129-146)
        run_case(outdir, "islands_72x36x8", FT, 72, 36, 8, 600.0, true; grid_type = :gaussian_islands)
        # the closure the reference keeps next to `nothing` (src/baroclinic_instability_model.jl:31); ν, κ large enough to matter
        run_case(outdir, "closure_128x64x8", FT, 128, 64, 8, 1200.0, true;
                 closure = VerticalScalarDiffusivity(VerticallyImplicitTimeDiscretization(), κ = 1e-3, ν = 1e-2))
        # src/baroclinic_instability_model.jl:30, sharding/less_simple_sharding_problem.jl:84-93
        run_case(outdir, "catke_128x64x8", FT, 128, 64, 8, 1200.0, true;
                 closure = Oceananigans.TurbulenceClosures.CATKEVerticalDiffusivity())
        run_data_free(outdir, FT, 8, 6)      # 48 x 24 x 6: the size of tests/golden/oracle_f64_coupled_48x24x6.npz
    end
end

main(ARGS)
