"""crowdsam_amd: MI355X-native (gfx950) implementation of Crowd-SAM's dense-prompt inference path."""
__version__ = "0.1.0"
