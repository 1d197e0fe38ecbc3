"""differt2d_amd -- MI355X-native drop-in for DiffeRT2d's power-map hot path."""

__version__ = "0.1.0"
