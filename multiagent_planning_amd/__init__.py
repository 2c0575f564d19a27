"""multiagent_planning_amd -- MI355X-native DMPC per-agent horizon-QP hot path.

Host-side mirror of the reference's MATLAB interface (`api`) over the C ABI of libdmpc_hip.so
(`_lib`).  See DESIGN.md / INTEGRATION.md at the repository root.
"""
from ._lib import Dmpc, DmpcError, DmpcParams, make_params, model_matrices, posvel_matrix, VARIANTS  # noqa: F401
from ._lib import ST_SOLVED, ST_OUTBOUND, ST_COLL, ST_INFEAS, ST_CAPACITY, ST_ITERCAP, ST_REACHED  # noqa: F401
