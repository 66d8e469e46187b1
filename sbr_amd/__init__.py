"""Importable alias of the `sequence-based-recommendations_amd/` package directory (a hyphen is
not importable): `import sbr_amd.engine` loads sequence-based-recommendations_amd/engine.py."""
import os

__path__ = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                         "sequence-based-recommendations_amd")]
