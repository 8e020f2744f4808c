"""mopa_amd: MI355X-native (gfx950) hot path of MoPA behind the reference's model-factory API."""
__version__ = "0.1.0"
