# bench/cpu_reference.jl -- the CPU baseline BASELINE.md section 3 names: the reference's own Julia CPU path.
#
#   julia --project=<GB-25 checkout> --threads=<cores> bench/cpu_reference.jl [Nx Ny Nz [Nsteps [dt]]]
#
# Uses only the GordonBell25 API (src/baroclinic_instability_model.jl:17-85, src/timestepping_utils.jl:21-45) on
# `CPU()` with Float32, default optimisation level (the reference's `-O0` is a compile-latency setting,
# sharding/alps_scaling_test.jl:85).  Config 1 of BASELINE.json is the default: 128x64x8, 100 AB2 steps.  The initial
# state is the one bench.py uses: set_baroclinic_instability! plus 1e-3 * U(0,1) velocity noise from the build-owned
# counter RNG (seed 42), so that the two programs time the same workload.  Prints one JSON line in the shape of
# bench.py's "cpu_baseline" object with kind = "reference".
#
# NOT EXECUTED in the build image (no Julia there).  Needs the pinned environment of the GB-25 checkout
# (Oceananigans =0.96.26, Project.toml:37).
using GordonBell25
using Oceananigans
using Printf

Oceananigans.defaults.FloatType = Float32

# SplitMix64 on the linear (column-major) index: tests/helpers.py counter_rng
function counter_rng(dims::NTuple{N,Int}, seed::Integer, salt::Integer) where {N}
    n = prod(dims)
    out = Array{Float64}(undef, dims)
    @inbounds for q in 0:n-1
        x = UInt64(q) + UInt64(seed) * 0x9E3779B97F4A7C15 + UInt64(salt) * 0xD1B54A32D192ED03
        x ⊻= x >> 30; x *= 0xBF58476D1CE4E5B9
        x ⊻= x >> 27; x *= 0x94D049BB133111EB
        x ⊻= x >> 31
        out[q + 1] = Float64(x >> 11) / Float64(UInt64(1) << 53)
    end
    return out
end

function main(args)
    Nx = length(args) >= 3 ? parse(Int, args[1]) : 128
    Ny = length(args) >= 3 ? parse(Int, args[2]) : 64
    Nz = length(args) >= 3 ? parse(Int, args[3]) : 8
    Nsteps = length(args) >= 4 ? parse(Int, args[4]) : 100
    Δt = length(args) >= 5 ? parse(Float64, args[5]) : 1200.0

    model = GordonBell25.baroclinic_instability_model(CPU(), Nx, Ny, Nz; Δt, halo = (8, 8, 8))
    GordonBell25.set_baroclinic_instability!(model)
    ui = 1e-3 .* counter_rng(size(model.velocities.u), 42, 1)
    vi = 1e-3 .* counter_rng(size(model.velocities.v), 42, 2)
    set!(model, u = ui, v = vi)

    GordonBell25.first_time_step!(model)
    GordonBell25.loop!(model, 2)                    # compile everything the timed loop runs
    t0 = time_ns()
    GordonBell25.loop!(model, Nsteps)
    elapsed = (time_ns() - t0) * 1e-9
    @printf("{\"value\": %.6g, \"unit\": \"steps/s\", \"cores\": %d, \"kind\": \"reference\", \"sample\": \"%d time steps of baroclinic_instability_model(CPU(), %d, %d, %d; dt=%g) Float32 after first_time_step!, Oceananigans %s\"}\n",
            Nsteps / elapsed, Threads.nthreads(), Nsteps, Nx, Ny, Nz, Δt, string(pkgversion(Oceananigans)))
end

main(ARGS)
