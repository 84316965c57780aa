"""Stand-in for torchrl 0.1.1 (see ../README.md)."""
__version__ = "0.1.1+hns.fake"
