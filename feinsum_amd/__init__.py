"""
feinsum_amd -- MI355X-native evaluation backend behind feinsum's builder and
measure API for the 3D DG-wave batched einsums (grad / div / face-mass).

The public names mirror ``feinsum/__init__.py`` of the reference for the hot
path (reference: ``src/feinsum/__init__.py:1-68``); everything loopy-, sqlite-
or autotuner-related is out of scope and absent.
"""

from feinsum_amd.cl_utils import FakeCLDevice
from feinsum_amd.contraction_schedule import (ContractionSchedule, count_ops,
                                              get_opt_einsum_contraction_schedule,
                                              get_trivial_contraction_schedule)
from feinsum_amd.diagnostics import (EinsumTunitMatchError, HipLibraryError, InvalidParameterError,
                                     NoDevicePeaksInfoError, NoFactInDatabaseError,
                                     TransformValidationError)
from feinsum_amd.einsum import (Array, BatchedEinsum, EinsumAxisAccess, FreeAxis, SizeParam,
                                SummationAxis)
from feinsum_amd.family import KernelPlan, match_family
from feinsum_amd.make_einsum import array, batched_einsum, einsum
from feinsum_amd.canonicalization import canonicalize_einsum
from feinsum_amd.sql_utils import (QueryInfo, get_timed_einsums_in_db, query, query_reference_archive, record_facts,
                                   retrieve)
from feinsum_amd.operator import BoundOperator, bind_operator, evaluate_operator
from feinsum_amd.measure import (DeviceQueue, evaluate, generate_input_arrays, generate_out_arrays,
                                 get_roofline_flop_rate, measure_giga_op_rate,
                                 stringify_comparison_vs_roofline, timeit, timeit_details,
                                 validate_batched_einsum_transform)

__all__ = (
    "Array", "BatchedEinsum", "BoundOperator", "ContractionSchedule", "DeviceQueue", "EinsumAxisAccess",
    "EinsumTunitMatchError", "FakeCLDevice", "FreeAxis", "HipLibraryError", "InvalidParameterError", "KernelPlan",
    "NoDevicePeaksInfoError", "NoFactInDatabaseError", "QueryInfo", "SizeParam", "SummationAxis",
    "TransformValidationError", "array", "batched_einsum", "bind_operator", "canonicalize_einsum", "count_ops", "einsum", "evaluate", "evaluate_operator",
    "generate_input_arrays", "generate_out_arrays", "get_opt_einsum_contraction_schedule",
    "get_roofline_flop_rate", "get_timed_einsums_in_db", "get_trivial_contraction_schedule", "match_family",
    "measure_giga_op_rate", "query", "query_reference_archive", "record_facts", "retrieve", "stringify_comparison_vs_roofline", "timeit", "timeit_details",
    "validate_batched_einsum_transform",
)

__version__ = "0.1.0"
