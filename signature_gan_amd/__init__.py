"""Import shim: the product lives in ``signature-gan_amd/`` (not an importable name); this
package simply points its search path there so ``import signature_gan_amd.engine`` works."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "signature-gan_amd")
__path__.insert(0, _real)
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _os, _f, _real
