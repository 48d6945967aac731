"""Import alias: the package directory is named `zero-tig_amd` (hyphen, per the build spec), which the `import` statement
cannot spell.  `import zerotig_amd` gives the same package object."""
import importlib
import sys

_pkg = importlib.import_module("zero-tig_amd")
sys.modules[__name__] = _pkg
