"""Stub: re-export the product's gin-lite reader under the name the reference imports."""
import importlib.util, os, sys
_p = os.path.join(os.path.dirname(__file__), "..", "..", "..", "..", "neural-waveshaping-synthesis_amd", "ginlite.py")
_spec = importlib.util.spec_from_file_location("_ginlite_for_reference", os.path.abspath(_p))
_m = importlib.util.module_from_spec(_spec)
sys.modules["_ginlite_for_reference"] = _m
_spec.loader.exec_module(_m)
configurable = _m.configurable
external_configurable = _m.external_configurable
config_scope = _m.config_scope
parse_config_file = _m.parse_config_file
parse_config = _m.parse_config
constant = _m.constant
clear_config = _m.clear_config
query_parameter = _m.query_parameter
REQUIRED = _m.REQUIRED
