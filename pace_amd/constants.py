"""Physical constants used on the acoustic path.

The reference selects a constant set with ``PACE_CONSTANTS`` (GFDL | GFS | GEOS)
[REF README.md:88-93]; the values themselves live in the un-vendored NDSL
submodule (``ndsl/constants.py``), so the numbers below are restated from the
public FV3 / NDSL sources.  Every kernel takes them as run-time parameters
(see ``include/fv3_mi355x.h: fv3_constants``) so a different set is a data change.
"""
import math
import os
from dataclasses import dataclass

N_HALO_DEFAULT = 3  # [REF driver/pace/driver/driver.py:21,177]

X_DIM = "x"
X_INTERFACE_DIM = "x_interface"
Y_DIM = "y"
Y_INTERFACE_DIM = "y_interface"
Z_DIM = "z"
Z_INTERFACE_DIM = "z_interface"


@dataclass(frozen=True)
class ConstantSet:
    name: str
    RADIUS: float
    OMEGA: float
    GRAV: float
    RDGAS: float
    RVGAS: float
    CP_AIR: float
    DZ_MIN: float
    PI: float = math.pi
    SECONDS_PER_DAY: float = 86400.0
    CNST_0P20: float = 0.20

    @property
    def RGRAV(self):
        return 1.0 / self.GRAV

    @property
    def KAPPA(self):
        return self.RDGAS / self.CP_AIR

    @property
    def CV_AIR(self):
        return self.CP_AIR - self.RDGAS

    @property
    def RDG(self):
        return -self.RDGAS / self.GRAV

    @property
    def ZVIR(self):
        return self.RVGAS / self.RDGAS - 1.0


GFS = ConstantSet(
    name="GFS",
    RADIUS=6.3712e6,
    OMEGA=7.2921e-5,
    GRAV=9.80665,
    RDGAS=287.05,
    RVGAS=461.50,
    CP_AIR=1004.6,
    DZ_MIN=2.0,
)
GFDL = ConstantSet(
    name="GFDL",
    RADIUS=6371.0e3,
    OMEGA=7.292e-5,
    GRAV=9.80,
    RDGAS=287.04,
    RVGAS=461.50,
    CP_AIR=287.04 / (2.0 / 7.0),
    DZ_MIN=2.0,
)
GEOS = ConstantSet(
    name="GEOS",
    RADIUS=6.371e6,
    OMEGA=2.0 * math.pi / 86164.0,
    GRAV=9.80665,
    RDGAS=8314.47 / 28.965,
    RVGAS=8314.47 / 18.015,
    CP_AIR=(8314.47 / 28.965) / (2.0 / 7.0),
    DZ_MIN=6.0,
)

_SETS = {"GFS": GFS, "GFDL": GFDL, "GEOS": GEOS}


def get_constants(name: str | None = None) -> ConstantSet:
    """Constant set named by ``PACE_CONSTANTS`` (default GFS)."""
    if name is None:
        name = os.environ.get("PACE_CONSTANTS", "GFS")
    try:
        return _SETS[name.upper()]
    except KeyError:
        raise ValueError(f"unknown PACE_CONSTANTS set {name!r}; expected one of {sorted(_SETS)}")


def float_precision_bits() -> int:
    """``PACE_FLOAT_PRECISION`` (32 | 64, default 64) [REF README.md:94]."""
    bits = int(os.environ.get("PACE_FLOAT_PRECISION", "64"))
    if bits not in (32, 64):
        raise ValueError("PACE_FLOAT_PRECISION must be 32 or 64")
    return bits
