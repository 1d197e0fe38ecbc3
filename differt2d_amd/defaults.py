"""Default values (same names and values as the reference's ``differt2d/defaults.py:3-15``)."""

DEFAULT_ALPHA: float = 100.0
"""Default ``alpha`` of :func:`differt2d_amd.logic.activation`."""

DEFAULT_PATCH: float = 0.0
"""Default patch applied to ``Interactable.intersects_cartesian``."""

DEFAULT_R_COEF: float = 0.5
"""Default real reflection coefficient."""

DEFAULT_HEIGHT: float = 0.1
"""Default TX antenna height used by :func:`differt2d_amd.utils.received_power`."""
