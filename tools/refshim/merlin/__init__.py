"""`merlin.MerlinTranscript` stand-in: re-exports the restated Merlin v1.0 from oracle/."""
import os, sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", "..")))
from oracle.strobe_merlin import MerlinTranscript  # noqa: E402,F401
