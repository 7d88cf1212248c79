"""MI355X-native `furnace/` tree: put THIS directory on sys.path where the
reference's config.py puts `<TorchSeg>/furnace` (config.py:49-54) and the
unchanged model/<family>/<exp>/{network,train}.py import from it."""
