"""TEST INFRASTRUCTURE: the oracle-backed stand-in for the rasterizer lives in oracle/cpu_render.py (bench.py's
cpu_baseline leg drives it through render() as well); re-exported here for the tests."""
from oracle.cpu_render import OracleRasterize, OracleRasterizer  # noqa: F401
