"""multipath-nn hot path, MI355X-native: operator surface (layer_types, net_types),
execution engine (_plan), C-ABI binding (_hip), data-parallel reducer (_dp) and
the harness pieces (data, desc)."""
