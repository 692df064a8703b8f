import sys, types
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import spblas_reference_amd as sp
sp.api._Handle._tls = types.SimpleNamespace()   # every thread sees the same table again: the old behaviour
import pytest
sys.exit(pytest.main(["tests/test_gpu_threads.py", "-q", "-x", "-p", "no:cacheprovider"]))
