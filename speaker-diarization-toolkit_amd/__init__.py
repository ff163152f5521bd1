"""MI355X-native speaker-embedding + assignment path for the CLIAI/speaker-diarization-toolkit
backend plug-in API.  See DESIGN.md (scope, kernels, layout) and INTEGRATION.md (how the
toolkit binds it).  The directory name contains a hyphen, so load it with
``importlib.import_module("speaker-diarization-toolkit_amd")`` (the toolkit's registry does exactly
that: speaker_detection_backends/base.py:291-292)."""
__version__ = "0.1.0"
BACKEND_NAME = "mi355x"
