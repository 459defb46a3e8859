"""Offline tools (reference tools/): presets / initial conditions and the frame recorder."""
