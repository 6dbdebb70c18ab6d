"""Import shim: `import qcnn_amd` loads the package that lives in
`quaternion-convolutional-neural-networks-for-end-to-end-automatic-speech-recognition_amd/`
(a directory name that is not a Python identifier)."""
import importlib.util
import os
import sys

_DIR = os.path.join(os.path.dirname(os.path.realpath(__file__)),
                    'quaternion-convolutional-neural-networks-for-end-to-end-automatic-speech-recognition_amd')
_spec = importlib.util.spec_from_file_location('qcnn_amd', os.path.join(_DIR, '__init__.py'),
                                               submodule_search_locations=[_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules['qcnn_amd'] = _mod
_spec.loader.exec_module(_mod)
