"""piccolo.jl_amd -- MI355X-native evaluator for the Pade collocation constraint of
Piccolo.jl / DirectTrajOpt.jl (one hot path; see DESIGN.md and SURVEY.md section 8).

The directory name contains a dot, so it is imported through the loader shim
``piccolo_jl_amd.py`` at the repository root:  ``import piccolo_jl_amd as pa``.
"""
from . import _lib, distributed, integrators, objectives, quantum, synthetic, trajectory
from ._lib import PclError, build_library
from .integrators import (
    BilinearIntegrator,
    DerivativeIntegrator,
    HipPadeIntegrator,
    HipPadeMemberIntegrator,
    HipPadeMultistart,
    eval_hessian_of_lagrangian,
    eval_jacobian,
    evaluate_,
    hessian_structure,
    jacobian_structure,
    unitary_rollout,
    unitary_rollout_fidelity,
)
from .objectives import (CoherentKetInfidelityObjective, DensityMatrixInfidelityObjective, DensityMatrixPureStateInfidelityObjective, EmbeddedOperator,
                         KetInfidelityObjective, Objective, QuadraticRegularizer, UnitaryInfidelityObjective, get_subspace_indices)
from .quantum import (
    GATES,
    PAULIS,
    CompositeQuantumSystem,
    MultiTransmonSystem,
    OpenQuantumSystem,
    QuantumSystem,
    TransmonDipoleCoupling,
    TransmonSystem,
    annihilate,
    compact_iso_to_density,
    density_to_compact_iso,
    iso,
    iso_vec_to_operator,
    lift_operator,
    operator_to_iso_vec,
)
from .trajectory import NamedTrajectory, add_control_derivatives, density_trajectory, ket_trajectory, sampling_trajectory, unitary_trajectory

__version__ = "0.4.0"
