"""Import the *Python reference* (vita-epfl/trajnetplusplusbaselines) from
/root/reference -- possible only in the build container, never on the GPU box.

Used by oracle/gen_golden.py (fixture generation) and by the optional
``tests/test_oracle_vs_reference.py`` (skipped when /root/reference is absent).
The reference's un-vendored third-party imports are satisfied with empty
module stubs; nothing on the LSTM / grid-pooling path calls into them
(SURVEY.md 8c).
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get('TRAJNET_REFERENCE_ROOT', '/root/reference')


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, 'trajnetbaselines'))


def import_reference():
    """Returns the reference ``trajnetbaselines.lstm`` package."""
    if not available():
        raise RuntimeError('reference checkout not found at %s' % REFERENCE_ROOT)
    for name in ['trajnetplusplustools', 'trajnetplusplustools.show', 'trajnetplusplustools.interactions',
                 'socialforce', 'socialforce.potentials', 'socialforce.field_of_view', 'rvo2', 'pykalman',
                 'pysparkling']:
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules['trajnetplusplustools.interactions'].collision_avoidance = None
    sys.modules['socialforce.potentials'].PedPedPotential = None
    sys.modules['socialforce.field_of_view'].FieldOfView = None
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import trajnetbaselines.lstm as ref_lstm  # noqa: E402
    return ref_lstm
