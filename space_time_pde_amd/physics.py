"""Rayleigh-Benard (RB2) governing equations as a PDELayer (mirrors experiments/rb2d/physics.py:6-64).

``get_rb2_pde_layer`` keeps the reference signature and equation names.  The residuals, with
nu_t = 1/t_crop, nu_x = 1/x_crop, nu_z = 1/z_crop, P = (Ra Pr)^-1/2, R = (Ra/Pr)^-1/2 (reference :19-24):
    transport_eqn_b: nu_t b_t - P (nu_x^2 b_xx + nu_z^2 b_zz)            + u nu_x b_x + w nu_z b_z
    transport_eqn_u: nu_t u_t - R (nu_x^2 u_xx + nu_z^2 u_zz) + p_x      + u nu_x u_x + w nu_z u_z
    transport_eqn_w: nu_t w_t - R (nu_x^2 w_xx + nu_z^2 w_zz) + p_z - b  + u nu_x w_x + w nu_z w_z
    continuity     : nu_x u_x + nu_z w_z                                   (optional)
Input variables are named 't, x, z' and are POSITIONAL (column 0, 1, 2 of the query points) exactly as in the
reference (quirk a-Q4: the data's axis order is (t, z, x), so the symbol x differentiates the data's z axis).
"""
from .pde import PDELayer

_OUT = ('p', 'b', 'u', 'w')


def _transport(q, nt, nx, nz, diffusivity, source):
    laplace = '(({nx})**2*dif(dif({q},x),x)+({nz})**2*dif(dif({q},z),z))'.format(nx=nx, nz=nz, q=q)
    advect = '(u*{nx}*dif({q},x)+w*{nz}*dif({q},z))'.format(nx=nx, nz=nz, q=q)
    return '{nt}*dif({q},t)-{k}*{lap}{src}+{adv}'.format(nt=nt, q=q, k=diffusivity, lap=laplace, src=source,
                                                          adv=advect)


def get_rb2_pde_layer(mean=None, std=None, t_crop=2., z_crop=1., x_crop=2., prandtl=1., rayleigh=1e6,
                      use_continuity=False):
    """PDE layer for the RB2 governing equations; forward method still has to be set by the caller.

    mean/std: per-channel (p, b, u, w) normalisation constants or None; when given, every channel v is replaced by
    v*std+mean inside and outside ``dif`` (change of variables, reference :38-56).
    """
    P = (rayleigh * prandtl) ** (-1 / 2)
    R = (rayleigh / prandtl) ** (-1 / 2)
    nt, nz, nx = 1. / t_crop, 1. / z_crop, 1. / x_crop
    equations = [
        ('transport_eqn_b', _transport('b', nt, nx, nz, P, '')),
        ('transport_eqn_u', _transport('u', nt, nx, nz, R, '+dif(p,x)')),
        ('transport_eqn_w', _transport('w', nt, nx, nz, R, '+dif(p,z)-b')),
    ]
    if use_continuity:
        equations.append(('continuity', '{nx}*dif(u,x)+{nz}*dif(w,z)'.format(nx=nx, nz=nz)))

    subs_dict = None
    if (mean is not None) or (std is not None):
        if (mean is None) or (std is None):
            raise ValueError('mean and std must either be both None, or both arrays of len 4.')
        if not (hasattr(mean, '__len__') and hasattr(std, '__len__')):
            raise TypeError('mean and std must be arrays of len 4. instead they are {} and {}'.format(
                type(mean), type(std)))
        if not (len(mean) == 4 and len(std) == 4):
            raise ValueError('mean and std must be arrays of len 4. instead they are of len {} and {}'.format(
                len(mean), len(std)))
        subs_dict = {v: '{}*{}+{}'.format(v, std[i], mean[i]) for i, v in enumerate(_OUT)}

    layer = PDELayer(in_vars='t, x, z', out_vars=', '.join(_OUT))
    for name, eqn in equations:
        layer.add_equation(eqn, name, subs_dict=subs_dict)
    return layer
