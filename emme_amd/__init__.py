"""emme_amd -- MI355X (gfx950) dispersion-matrix assembly + eigenvalue search.

Thin Python face of the C ABI in include/emme_hip.h (built by emme_amd/csrc/Makefile into
emme_amd/libemme_hip.so).  There is no CPU fallback: every compute entry point raises
if the HIP library or a gfx950 device is missing.
"""
from ._lib import (EmmeError, Params, Profile, Context, params_from_json, params_from_dict,
                   tables, weight, lib_path, load, json_text, null_vector, scan_values, run_json,
                   release_pooled_memory, Comm, comm_unique_id, bessel, Options, default_options,
                   set_default_options, FILL_AUTO, FILL_UNION, FILL_LANES, comm_available,
                   gather_pack, gather_unpack)

__all__ = ["EmmeError", "Params", "Profile", "Context", "params_from_json", "params_from_dict",
           "tables", "weight", "lib_path", "load", "json_text", "null_vector", "scan_values", "run_json",
           "release_pooled_memory", "Comm", "comm_unique_id", "bessel", "Options", "default_options",
           "set_default_options", "FILL_AUTO", "FILL_UNION", "FILL_LANES", "comm_available", "gather_pack",
           "gather_unpack"]
