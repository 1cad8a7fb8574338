"""space_time_pde_amd -- MI355X-native MeshfreeFlowNet hot path.

Drop-in for the reference's flat modules (``pde``, ``local_implicit_grid``, ``regular_nd_grid_interpolation``,
``implicit_net``, ``nonlinearities``, ``unet3d``, ``physics``): either ``from space_time_pde_amd import pde``, or
``sys.path.append("<repo>/space_time_pde_amd/flat")`` and ``import pde`` exactly like
experiments/rb2d/train.py:19-27 does with ``../../src`` (the flat names alias the package modules).
"""

__version__ = "0.1.0"
