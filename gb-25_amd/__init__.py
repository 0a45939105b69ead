"""gb25_amd -- MI355X-native implementation of the time-step hot path of PRONTOLab/GB-25.

Public names mirror `GordonBell25` (GB-25 src/GordonBell25.jl:3-4 and the un-exported helpers
its scripts call); Julia's trailing `!` is dropped.  Kernels live in libgb25hip.so behind the
C ABI of include/gb25.h; this package only sequences calls and moves host arrays.
"""
from .binding import GB25Error, HipBackend, LIB_PATH, load_library
from .build import build_library
from .correctness import approx_equal, compare_states, sync_states
from .model import (CATKEVerticalDiffusivity, default_ocean_closure, Field, HydrostaticFreeSurfaceModel, VerticalScalarDiffusivity,
                    baroclinic_instability_model, first_time_step, initialize,
                    loop, resolution_to_points, set_baroclinic_instability, set_top_flux, time_step, update_state,
                    tupled_fill_halo_regions_workload, compute_tendencies_workload,
                    compute_boundary_tendencies_workload, compute_interior_momentum_tendencies_workload,
                    compute_interior_tracer_tendencies_workload, compute_auxiliaries_workload,
                    fill_halo_regions_workload, ab2_step_workload,
                    correct_velocities_and_cache_previous_tendencies_workload)
from .data_free import (PrescribedAtmosphere, analytic_atmosphere, data_free_ocean_climate_model_init,
                        set_prescribed_atmosphere, set_data_free_state, zonal_wind, sunlight, Tatm)
from .sharding import factors
from .arg_parsing import (float_type_from_args, float_type_from_string, interior_size, multifloat_from_args,
                          parse_baroclinic_instability_args)
from .sharded_io import load_all_fields, load_global_field, save_model_state


class GPU:
    """Architecture object: `baroclinic_instability_model(GPU(), Nx, Ny, Nz; dt=...)`
    (the reference passes CPU() / GPU() / ReactantState(), correctness/..._run.jl:33-34)."""

    def __init__(self, device=0, float_type="Float32"):
        self.device = device
        self.float_type = float_type   # "Float32" | "Float64" (float_type_from_args)

    def __call__(self, Nx, Ny, Nz, **kw):
        kw.setdefault("device", self.device)
        kw.setdefault("float_type", self.float_type)
        return HipBackend(Nx, Ny, Nz, **kw)


__all__ = [n for n in dir() if not n.startswith("_")]
