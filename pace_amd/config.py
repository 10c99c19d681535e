"""Configuration of the acoustic path.

Field names are the Fortran-namelist names the reference's ``dycore_config``
uses [REF driver/examples/configs/baroclinic_c12.yaml:43-93; driver.py:257-263];
the sub-config split mirrors ``pyFV3._config`` (AcousticDynamicsConfig,
DGridShallowWaterLagrangianDynamicsConfig, RiemannConfig)
[REF tests/main/fv3core/test_config.py:10-16].  Same-named fields keep the
same type across the classes [REF test_config.py:80-87].
"""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass
from typing import Tuple


@dataclass
class RiemannConfig:
    p_fac: float = 0.05
    a_imp: float = 1.0
    use_logp: bool = False
    beta: float = 0.0


@dataclass
class DGridShallowWaterLagrangianDynamicsConfig:
    dddmp: float = 0.5
    d2_bg: float = 0.0
    d2_bg_k1: float = 0.2
    d2_bg_k2: float = 0.1
    d4_bg: float = 0.15
    ke_bg: float = 0.0
    nord: int = 3
    n_sponge: int = 48
    grid_type: int = 0
    d_ext: float = 0.0
    hord_dp: int = 6
    hord_tm: int = 6
    hord_mt: int = 6
    hord_vt: int = 6
    do_f3d: bool = False
    do_skeb: bool = False
    d_con: float = 1.0
    vtdm4: float = 0.06
    inline_q: bool = False
    convert_ke: bool = False
    do_vort_damp: bool = True
    hydrostatic: bool = False


@dataclass
class AcousticDynamicsConfig:
    """Everything ``AcousticDynamics`` reads.  Defaults = every reference yaml."""

    npx: int = 13
    npy: int = 13
    npz: int = 79
    layout: Tuple[int, int] = (1, 1)
    ntiles: int = 6
    dt_atmos: float = 225.0
    k_split: int = 1
    n_split: int = 1
    # dynamics
    a_imp: float = 1.0
    beta: float = 0.0
    p_fac: float = 0.05
    use_logp: bool = False
    hydrostatic: bool = False
    grid_type: int = 0
    nested: bool = False
    stretched_grid: bool = False
    # transport
    hord_dp: int = 6
    hord_mt: int = 6
    hord_tm: int = 6
    hord_vt: int = 6
    hord_tr: int = 8
    # damping
    nord: int = 3
    d2_bg: float = 0.0
    d2_bg_k1: float = 0.2
    d2_bg_k2: float = 0.1
    d4_bg: float = 0.15
    dddmp: float = 0.5
    d_con: float = 1.0
    d_ext: float = 0.0
    delt_max: float = 0.002
    ke_bg: float = 0.0
    vtdm4: float = 0.06
    do_vort_damp: bool = True
    n_sponge: int = 48
    convert_ke: bool = False
    do_skeb: bool = False
    do_f3d: bool = False
    inline_q: bool = False
    # Rayleigh damping
    rf_fast: bool = True
    rf_cutoff: float = 3000.0
    tau: float = 10.0
    use_old_omega: bool = True
    breed_vortex_inline: bool = False

    def validate(self):
        """The kernels are specialised like the reference configs; anything else fails loudly."""
        bad = []
        if self.hydrostatic:
            bad.append("hydrostatic=True")
        if self.a_imp <= 0.999:
            bad.append("a_imp<=0.999 (only SIM1)")
        if self.beta != 0.0:
            bad.append("beta!=0")
        if self.d_ext != 0.0:
            bad.append("d_ext!=0")
        if self.use_logp:
            bad.append("use_logp")
        if self.grid_type != 0:
            bad.append("grid_type!=0")
        if self.nested or self.stretched_grid:
            bad.append("nested/stretched grid")
        for h in ("hord_dp", "hord_mt", "hord_tm", "hord_vt"):
            if getattr(self, h) not in (5, 6):
                bad.append(f"{h}={getattr(self, h)} (5 or 6)")
        if not (0 <= self.nord <= 3):
            bad.append("nord outside 0..3")
        if self.do_skeb or self.do_f3d or self.inline_q:
            bad.append("do_skeb/do_f3d/inline_q")
        if bad:
            raise NotImplementedError("unsupported dycore_config on the MI355X acoustic path: " + ", ".join(bad))
        return self

    @property
    def riemann(self) -> RiemannConfig:
        return RiemannConfig(p_fac=self.p_fac, a_imp=self.a_imp, use_logp=self.use_logp, beta=self.beta)

    @property
    def d_grid_shallow_water(self) -> DGridShallowWaterLagrangianDynamicsConfig:
        names = {f.name for f in dataclasses.fields(DGridShallowWaterLagrangianDynamicsConfig)}
        return DGridShallowWaterLagrangianDynamicsConfig(**{n: getattr(self, n) for n in names})

    @classmethod
    def from_dict(cls, d: dict) -> "AcousticDynamicsConfig":
        names = {f.name for f in dataclasses.fields(cls)}
        kw = {k: v for k, v in d.items() if k in names}
        if "layout" in kw:
            kw["layout"] = tuple(kw["layout"])
        return cls(**kw)
