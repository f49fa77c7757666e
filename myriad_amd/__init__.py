"""myriad_amd -- MI355X-native batched trajectory-optimisation engine with the API surface of
nikihowe/myriad's trajectory-optimisation path (see DESIGN.md / INTEGRATION.md)."""
__version__ = "0.1.0"
