"""matryodshka_amd -- MI355X (gfx950) native multi-sphere-image infer -> render path.

    from matryodshka_amd import MSI          # needs libmsi_hip.so (python -m matryodshka_amd.build)

The package import itself stays light so that `python -m matryodshka_amd.build`
works before the shared library exists; touching `MSI`, `nets` or `_native`
loads the library and raises if it is missing (there is no CPU fallback).
"""
__version__ = "0.1.0"


def __getattr__(name):
    if name == "MSI":
        from .msi import MSI
        return MSI
    raise AttributeError(name)
