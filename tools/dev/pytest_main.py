"""Development only: pytest under tools/dev/with_lib.py (a variant library):  python tools/dev/with_lib.py <so> tools/dev/pytest_main.py tests/... -k ..."""
import sys

import pytest

sys.exit(pytest.main(sys.argv[1:]))
