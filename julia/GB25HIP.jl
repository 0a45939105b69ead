# GB25HIP.jl -- Julia binding of libgb25hip.so (the C ABI of include/gb25.h).
#
# The host-language side of the drop-in boundary: GB-25 is a Julia package (src/GordonBell25.jl) and this module is
# what its maintainers would `include` to run the HydrostaticFreeSurfaceModel time-step loop on MI355X through the
# hand-written HIP library instead of through Reactant/XLA.  It keeps the GordonBell25 entry points
# (first_time_step!, time_step!, loop! -- src/timestepping_utils.jl:21-45) and the per-phase workloads
# (src/precompile.jl:31-127) as one `ccall` each.  No KernelAbstractions, no AMDGPU.jl code generation: the kernels are
# prebuilt; AMDGPU.jl is only needed if zero-copy `ROCArray` views of the fields are wanted (see `device_view`).
#
# STATUS: written against include/gb25.h without a Julia installation at hand (none exists in the build image or on
# the GPU box); it has not been executed.  The ctypes binding gb-25_amd/binding.py is the one the test-suite runs, and
# this file mirrors it call for call.
module GB25HIP

export Config, Model, create, destroy!, first_time_step!, time_step!, loop!, initialize!, update_state!,
       fill_halo_regions!, compute_auxiliaries!, compute_tendencies!, ab2_step!, mask_immersed_fields!,
       correct_velocities_and_cache_previous_tendencies!, set_baroclinic_instability!, synchronize,
       parent_array, interior_array, set_parent!, set_interior!, clock, set_dt!, set_option!, get_option,
       comm_unique_id, comm_init_rccl!, comm_finalize!, set_top_flux!, set_bottom_height!, set_vertical_diffusivity!, set_closure_catke!, CatkeParameters, default_catke_parameters, set_catke_parameters!, set_bottom_drag!, set_tracer_advection_order!, set_prescribed_atmosphere!, compute_atmosphere_ocean_fluxes!, metric2, FIELD, OPTION, METRIC2

# One library per Oceananigans float type (src/arg_parsing.jl:12-16): Float32 -> libgb25hip.so, Float64 ->
# libgb25hip_f64.so; same symbols, gb25_real_bytes() tells them apart.
const LIBS = Dict(Float32 => get(ENV, "GB25HIP_LIB", "libgb25hip.so"),
                  Float64 => get(ENV, "GB25HIP_LIB_F64", "libgb25hip_f64.so"))

# gb25_field (include/gb25.h)
const FIELD = (u = 0, v = 1, w = 2, T = 3, S = 4, pHY = 5,
               Gn_u = 6, Gn_v = 7, Gn_T = 8, Gn_S = 9, Gm_u = 10, Gm_v = 11, Gm_T = 12, Gm_S = 13,
               eta = 14, U = 15, V = 16, eta_bar = 17, U_bar = 18, V_bar = 19, Gn_U = 20, Gn_V = 21,
               # closure = CATKEVerticalDiffusivity() (exist after set_closure_catke!)
               e = 22, Gn_e = 23, Gm_e = 24, κu = 25, κc = 26, κe = 27, Le = 28, Jb = 29,
               # diffusivity_fields.previous_velocities (u, v at the previous compute_diffusivities!)
               previous_u = 30, previous_v = 31)
# gb25_option
const OPTION = (kernels = 0, ab2_lookahead = 1, subcycle_lookahead = 2, subcycle_block = 3, fill_fused = 4,
                two_streams = 5, store_pressure = 6, split_tendencies = 7, pressure_precision = 8,
                immersed_kernels = 9, fold_fills = 10, lazy_corrector = 11, momentum_chunk_levels = 12, tracer_chunk_levels = 13,
                tracers_first = 14, w_on_the_fly = 15, sub_stream_priority = 16, subcycle_whole = 17, early_strips = 18,
                # restatement choices a Julia dump settles (DESIGN.md section 0), and two run-time knobs
                catke_stale_e_halos = 19, comm_timeout_seconds = 20, roctx_ranges = 21, substep_order = 22, fold_pivot_slaved = 23)

# mirror of gb25_config; isbits, passed by reference
Base.@kwdef mutable struct Config
    Nx::Int32 = 0
    Ny::Int32 = 0
    Nz::Int32 = 0
    halo::Int32 = 8                 # halo = (8, 8, 8) in every GB-25 script
    substeps::Int32 = 30            # SplitExplicitFreeSurface(substeps=30), src/baroclinic_instability_model.jl:22
    rank::Int32 = 0
    nranks::Int32 = 1
    device::Int32 = 0
    dt::Float64 = 60.0              # model.clock.last_Δt, src/baroclinic_instability_model.jl:82
    chi::Float64 = 0.1
    lat_south::Float64 = -80.0
    lat_north::Float64 = 80.0
    lon_west::Float64 = 0.0
    lon_east::Float64 = 360.0
    depth::Float64 = 4000.0
    zexp_h::Float64 = 30.0
    g::Float64 = 9.80665
    Omega::Float64 = 7.292115e-5
    radius::Float64 = 6371e3
    rho0::Float64 = 1020.0
    slab_mode::Int32 = 0
    grid_type::Int32 = 0            # gb25_grid_type: 0 = :simple_lat_lon, 1 = the Gaussian islands on the lat-lon grid,
                                    # 3 = TripolarGrid, 4 = :gaussian_islands (TripolarGrid + GridFittedBottom)
    ranks_y::Int32 = 1              # Partition(Rx, Ry, 1) (sharding/sharded_baroclinic_instability_simulation_run.jl:65-72):
                                    # Ry; nranks = Rx Ry, rank = ry Rx + rx; 1: x slabs
end

mutable struct Model{FT}
    ptr::Ptr{Cvoid}
    lib::String
    cfg::Config
end

struct GB25Error <: Exception
    msg::String
end
Base.showerror(io::IO, e::GB25Error) = print(io, "GB25Error: ", e.msg)

function check(m::Model, status::Integer, what::AbstractString)
    status == 0 && return nothing
    msg = unsafe_string(ccall((:gb25_last_error_string, m.lib), Cstring, (Ptr{Cvoid},), m.ptr))
    throw(GB25Error("$what failed with status $status: $msg"))   # a Julia exception on the Julia side of the ABI only
end

"""
    create(FT, cfg) -> Model{FT}

`baroclinic_instability_model(arch, Nx, Ny, Nz; Δt, halo, free_surface=SplitExplicitFreeSurface(substeps=...))`
(src/baroclinic_instability_model.jl:17-85): builds the grid metrics, allocates every field zeroed on the device.
"""
function create(::Type{FT}, cfg::Config) where {FT<:Union{Float32,Float64}}
    lib = LIBS[FT]
    nbytes = ccall((:gb25_real_bytes, lib), Int32, ())
    nbytes == sizeof(FT) || throw(GB25Error("$lib holds $(nbytes)-byte elements, expected $FT"))
    # the mirror of gb25_config must be the library's struct, field for field (a field added there must not shift silently here)
    cbytes = ccall((:gb25_config_bytes, lib), Int32, ())
    cbytes == sizeof(Config) || throw(GB25Error("gb25_config is $cbytes bytes in $lib and $(sizeof(Config)) in this binding"))
    out = Ref{Ptr{Cvoid}}(C_NULL)
    st = ccall((:gb25_create, lib), Cint, (Ref{Config}, Ref{Ptr{Cvoid}}), cfg, out)
    m = Model{FT}(out[], lib, cfg)
    if st != 0
        msg = m.ptr == C_NULL ? "gb25_create failed" :
              unsafe_string(ccall((:gb25_last_error_string, lib), Cstring, (Ptr{Cvoid},), m.ptr))
        m.ptr == C_NULL || ccall((:gb25_destroy, lib), Cvoid, (Ptr{Cvoid},), m.ptr)
        throw(GB25Error("gb25_create: status $st: $msg"))
    end
    finalizer(destroy!, m)
    return m
end

"""
    create(FT, Nx, Ny, Nz; Δt, halo=8, substeps=30, kw...)
"""
function create(::Type{FT}, Nx::Integer, Ny::Integer, Nz::Integer; Δt, halo = 8, substeps = 30, kw...) where {FT}
    cfg = Config(; Nx = Nx, Ny = Ny, Nz = Nz, dt = Δt, halo = halo, substeps = substeps, kw...)
    return create(FT, cfg)
end

function destroy!(m::Model)
    if m.ptr != C_NULL
        ccall((:gb25_destroy, m.lib), Cvoid, (Ptr{Cvoid},), m.ptr)
        m.ptr = C_NULL
    end
    return nothing
end

# ---- the phases (src/precompile.jl:31-42) and the composites (src/timestepping_utils.jl:21-45): one ccall each
for (jl, c) in ((:first_time_step!, :gb25_first_time_step),
                (:time_step!, :gb25_time_step),
                (:initialize!, :gb25_initialize),
                (:update_state!, :gb25_update_state),
                (:mask_immersed_fields!, :gb25_mask_immersed_fields),
                (:fill_halo_regions!, :gb25_fill_halo_regions),
                (:compute_auxiliaries!, :gb25_compute_auxiliaries),
                (:fill_diffusivity_halos!, :gb25_fill_diffusivity_halos),
                (:compute_momentum_tendencies!, :gb25_compute_momentum_tendencies),
                (:compute_tracer_tendencies!, :gb25_compute_tracer_tendencies),
                (:compute_boundary_tendencies!, :gb25_compute_boundary_tendencies),
                (:compute_tendencies!, :gb25_compute_tendencies),
                (:set_baroclinic_instability!, :gb25_set_baroclinic_instability),
                (:synchronize, :gb25_synchronize),
                (:comm_finalize!, :gb25_comm_finalize))
    @eval $jl(m::Model) = check(m, ccall(($(QuoteNode(c)), m.lib), Cint, (Ptr{Cvoid},), m.ptr), $(string(c)))
end

loop!(m::Model, Ninner::Integer) =
    check(m, ccall((:gb25_loop, m.lib), Cint, (Ptr{Cvoid}, Int32), m.ptr, Ninner), "gb25_loop")
ab2_step!(m::Model, Δt::Real, euler::Bool = false) =
    check(m, ccall((:gb25_ab2_step, m.lib), Cint, (Ptr{Cvoid}, Float64, Cint), m.ptr, Δt, euler), "gb25_ab2_step")
correct_velocities_and_cache_previous_tendencies!(m::Model, Δt::Real = 0.0) =
    check(m, ccall((:gb25_correct_velocities_and_cache_previous_tendencies, m.lib), Cint, (Ptr{Cvoid}, Float64),
                   m.ptr, Δt), "gb25_correct_velocities_and_cache_previous_tendencies")

# ---- FluxBoundaryCondition at the top of u, v, T, S (what ocean_simulation's coupled fluxes fill every step;
# src/data_free_ocean_climate_model.jl:26, src/precompile.jl:52-61): J at the interior points, positive upward
function set_top_flux!(m::Model{FT}, field::Integer, J::Union{Nothing, AbstractMatrix}) where FT
    a = J === nothing ? nothing : convert(Matrix{FT}, J)          # values in the model's float type
    p = a === nothing ? Ptr{Cvoid}(C_NULL) : Ptr{Cvoid}(pointer(a))
    GC.@preserve a check(m, ccall((:gb25_set_top_flux, m.lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}), m.ptr, field, p),
                         "gb25_set_top_flux")
end
# closure = VerticalScalarDiffusivity(VerticallyImplicitTimeDiscretization(), κ, ν) (src/baroclinic_instability_model.jl:31);
# ν = κ = 0: closure = nothing
set_vertical_diffusivity!(m::Model; ν::Real = 0, κ::Real = 0) =
    check(m, ccall((:gb25_set_vertical_diffusivity, m.lib), Cint, (Ptr{Cvoid}, Float64, Float64), m.ptr, ν, κ),
          "gb25_set_vertical_diffusivity")
# closure = CATKEVerticalDiffusivity() (src/baroclinic_instability_model.jl:30): tracers = (:T, :S, :e), diffusivity fields
# κu, κc, κe, Le, Jb; once after creation, before the initial state is set
set_closure_catke!(m::Model, on::Bool = true) =
    check(m, ccall((:gb25_set_closure_catke, m.lib), Cint, (Ptr{Cvoid}, Int32), m.ptr, on), "gb25_set_closure_catke")
# data-free forcing (src/data_free_ocean_climate_model.jl:12-70): one field of the PrescribedAtmosphere at the ocean's cell
# centres, halo cells included ((Nx + 2H) x (Ny + 2H) Float64); all seven set => coupled: first_time_step! / loop! compute the
# similarity-theory fluxes after every step and keep them in the top flux boundary conditions
const ATMOSPHERE = (u = 0, v = 1, T = 2, q = 3, p = 4, shortwave = 5, longwave = 6)
set_prescribed_atmosphere!(m::Model, field::Symbol, values::AbstractMatrix) =
    check(m, ccall((:gb25_set_prescribed_atmosphere, m.lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}), m.ptr,
                   getproperty(ATMOSPHERE, field), convert(Matrix{Float64}, values)), "gb25_set_prescribed_atmosphere")
compute_atmosphere_ocean_fluxes!(m::Model) =
    check(m, ccall((:gb25_compute_atmosphere_ocean_fluxes, m.lib), Cint, (Ptr{Cvoid},), m.ptr), "gb25_compute_atmosphere_ocean_fluxes")
# tracer_advection = WENO(order = 5) (default) | WENO(order = 7) (ClimaOcean's ocean_simulation)
set_tracer_advection_order!(m::Model, order::Integer) =
    check(m, ccall((:gb25_set_tracer_advection_order, m.lib), Cint, (Ptr{Cvoid}, Int32), m.ptr, order), "gb25_set_tracer_advection_order")
# quadratic bottom drag (ClimaOcean's ocean_simulation: bottom_drag_coefficient = 0.003); 0: none
set_bottom_drag!(m::Model, Cd::Real) =
    check(m, ccall((:gb25_set_bottom_drag, m.lib), Cint, (Ptr{Cvoid}, Float64), m.ptr, Cd), "gb25_set_bottom_drag")
# mirror of gb25_catke_parameters (isbits; the arrays are psi = u, c, e, D); defaults: default_catke_parameters()
struct CatkeParameters
    Cs::Float64; Cb::Float64; Csp::Float64; CRid::Float64; CRi0::Float64
    Chi::NTuple{4, Float64}; Clo::NTuple{4, Float64}; Cun::NTuple{4, Float64}; Cc::NTuple{4, Float64}; Ce::NTuple{4, Float64}
    CWu::Float64; CWw::Float64
    minimum_tke::Float64; minimum_convective_buoyancy_flux::Float64; negative_tke_damping_time_scale::Float64
    CWeps::Float64      # bottom TKE flux coefficient (CATKEEquation's Cᵂϵ)
end
function default_catke_parameters(lib)
    p = Ref{CatkeParameters}()
    ccall((:gb25_default_catke_parameters, lib), Cvoid, (Ptr{CatkeParameters},), p)
    return p[]
end
set_catke_parameters!(m::Model, p::CatkeParameters) =
    check(m, ccall((:gb25_set_catke_parameters, m.lib), Cint, (Ptr{Cvoid}, Ptr{CatkeParameters}), m.ptr, Ref(p)),
          "gb25_set_catke_parameters")
# GridFittedBottom(bottom_height): heights at the GLOBAL cell centres, (Nx_global, Ny) (every rank passes the same array)
set_bottom_height!(m::Model, zb::AbstractMatrix) =
    check(m, ccall((:gb25_set_bottom_height, m.lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), m.ptr, convert(Matrix{Float64}, zb)),
          "gb25_set_bottom_height")

# ---- the HOST's grid: the model steps on Oceananigans' own numbers, not on the library's stand-in generators.
# `grid` is the underlying OrthogonalSphericalShellGrid (TripolarGrid(arch; size, halo, z), src/model_utils.jl:134-137) built on
# CPU() over the GLOBAL domain; the 14 arrays go in gb25_metric2 order as the parents of grid.Δxᶠᶜᵃ ..., (Nx + 2H, Ny + 2H).
function set_curvilinear_grid!(m::Model, grid; Ω = 7.292115e-5)
    P(name) = convert(Matrix{Float64}, collect(parent(getproperty(grid, name)))[:, :, 1])
    fff = 2Ω .* sind.(P(:φᶠᶠᵃ))                                   # HydrostaticSphericalCoriolis at (Face, Face)
    arrays = [P(:Δxᶠᶜᵃ), P(:Δxᶜᶜᵃ), P(:Δxᶜᶠᵃ), P(:Δxᶠᶠᵃ), P(:Δyᶠᶜᵃ), P(:Δyᶜᶜᵃ), P(:Δyᶜᶠᵃ), P(:Δyᶠᶠᵃ),
              P(:Azᶜᶜᵃ), P(:Azᶠᶜᵃ), P(:Azᶜᶠᵃ), P(:Azᶠᶠᵃ), fff, P(:φᶜᶜᵃ)]
    nx, ny = size(arrays[1])
    ptrs = [pointer(a) for a in arrays]
    GC.@preserve arrays check(m, ccall((:gb25_set_curvilinear_grid, m.lib), Cint, (Ptr{Cvoid}, Ptr{Ptr{Float64}}, Int32, Int32),
                                       m.ptr, ptrs, nx, ny), "gb25_set_curvilinear_grid")
end
# grid.z faces, bottom to top: exponential_z_faces(Nz, depth) in the reference (src/model_utils.jl:56-62)
function set_vertical_faces!(m::Model, grid)
    Nz, Hz = grid.Nz, grid.Hz
    zf = convert(Vector{Float64}, collect(parent(grid.z.cᵃᵃᶠ))[Hz + 1:Hz + Nz + 1])
    check(m, ccall((:gb25_set_vertical_faces, m.lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int32), m.ptr, zf, length(zf)),
          "gb25_set_vertical_faces")
end
# everything an ImmersedBoundaryGrid(TripolarGrid, GridFittedBottom) holds (src/model_utils.jl:129-146), in one call
function set_grid!(m::Model, ibg)
    grid = ibg.underlying_grid
    set_curvilinear_grid!(m, grid)
    set_vertical_faces!(m, grid)
    Hx, Hy = grid.Hx, grid.Hy
    zb = collect(parent(ibg.immersed_boundary.bottom_height))[Hx + 1:Hx + grid.Nx, Hy + 1:Hy + grid.Ny, 1]
    set_bottom_height!(m, zb)
end
# gb25_metric2: horizontal metrics of an orthogonal curvilinear grid (grid_type >= 2) by location
const METRIC2 = (dxfc = 0, dxcc = 1, dxcf = 2, dxff = 3, dyfc = 4, dycc = 5, dycf = 6, dyff = 7,
                 azcc = 8, azfc = 9, azcf = 10, azff = 11, fff = 12, phicc = 13)
function metric2(m::Model, id::Integer, Nx::Integer, Ny::Integer, H::Integer)
    a = Matrix{Float64}(undef, Nx + 2H, Ny + 2H + 1)        # parent layout of a (Center, Face) field
    check(m, ccall((:gb25_get_metric2, m.lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Int64), m.ptr, id, a, length(a)),
          "gb25_get_metric2")
    return a
end

set_option!(m::Model, opt::Integer, value::Integer) =
    check(m, ccall((:gb25_set_option, m.lib), Cint, (Ptr{Cvoid}, Cint, Int32), m.ptr, opt, value), "gb25_set_option")
function get_option(m::Model, opt::Integer)
    v = Ref{Int32}(0)
    check(m, ccall((:gb25_get_option, m.lib), Cint, (Ptr{Cvoid}, Cint, Ref{Int32}), m.ptr, opt, v), "gb25_get_option")
    return v[]
end

# ---- clock: model.clock (src/model_utils.jl:150-155)
function clock(m::Model)
    t = Ref{Float64}(0); it = Ref{Int64}(0); dt = Ref{Float64}(0)
    check(m, ccall((:gb25_get_clock, m.lib), Cint, (Ptr{Cvoid}, Ref{Float64}, Ref{Int64}, Ref{Float64}),
                   m.ptr, t, it, dt), "gb25_get_clock")
    return (time = t[], iteration = it[], last_Δt = dt[])
end
set_dt!(m::Model, Δt::Real) =
    check(m, ccall((:gb25_set_dt, m.lib), Cint, (Ptr{Cvoid}, Float64), m.ptr, Δt), "gb25_set_dt")

# ---- fields: parent(field) / interior(field) as host Arrays (copies), column-major [i, j, k] like Oceananigans
function field_dims(m::Model, field::Integer, include_halos::Bool)
    d = zeros(Int32, 3)
    check(m, ccall((:gb25_field_dims, m.lib), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Int32}),
                   m.ptr, field, include_halos, d), "gb25_field_dims")
    return (Int(d[1]), Int(d[2]), Int(d[3]))
end
function _get(m::Model{FT}, field::Integer, include_halos::Bool) where {FT}
    a = Array{FT}(undef, field_dims(m, field, include_halos)...)
    check(m, ccall((:gb25_get_field, m.lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint),
                   m.ptr, field, a, include_halos), "gb25_get_field")
    return a
end
function _set!(m::Model{FT}, field::Integer, a::AbstractArray, include_halos::Bool) where {FT}
    dims = field_dims(m, field, include_halos)
    b = Array{FT}(reshape(a, dims))       # contiguous, converted to the library's element type
    check(m, ccall((:gb25_set_field, m.lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint),
                   m.ptr, field, b, include_halos), "gb25_set_field")
    return nothing
end
parent_array(m::Model, field::Integer) = _get(m, field, true)
interior_array(m::Model, field::Integer) = _get(m, field, false)
set_parent!(m::Model, field::Integer, a::AbstractArray) = _set!(m, field, a, true)
set_interior!(m::Model, field::Integer, a::AbstractArray) = _set!(m, field, a, false)

"""
    device_pointer(m, field) -> Ptr{FT}

Device address of `parent(field)` (exactly the Oceananigans parent layout) for zero-copy wrapping, e.g. with AMDGPU.jl
`unsafe_wrap(ROCArray, device_pointer(m, f), field_dims(m, f, true))`.  Handing out the pointer of u, v, T, S or of a
tendency pins those fields to the buffers handed out and turns the AB2 look-ahead off for this model (gb25.h).
"""
function device_pointer(m::Model{FT}, field::Integer) where {FT}
    p = Ref{Ptr{Cvoid}}(C_NULL)
    check(m, ccall((:gb25_field_device_ptr, m.lib), Cint, (Ptr{Cvoid}, Cint, Ref{Ptr{Cvoid}}), m.ptr, field, p),
          "gb25_field_device_ptr")
    return Ptr{FT}(p[])
end

# ---- x-slab decomposition: one Julia process per GPU (Config(rank=..., nranks=...)), halo exchange inside the library
"128 bytes of ncclGetUniqueId; rank 0 calls this and MPI.Bcast's the buffer."
function comm_unique_id(::Type{FT} = Float32) where {FT}
    id = zeros(UInt8, 128)
    st = ccall((:gb25_comm_unique_id, LIBS[FT]), Cint, (Ptr{UInt8},), id)
    st == 0 || throw(GB25Error("gb25_comm_unique_id: status $st (librccl could not be loaded?)"))
    return id
end
comm_init_rccl!(m::Model, id::Vector{UInt8}) =
    check(m, ccall((:gb25_comm_init_rccl, m.lib), Cint, (Ptr{Cvoid}, Ptr{UInt8}), m.ptr, id), "gb25_comm_init_rccl")

end # module
