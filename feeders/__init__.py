"""``feeders`` -- the package path the reference configs name (``feeder: feeders.feeder.Feeder``).
The implementation lives in ``2s-agcn_amd/feeders/``."""
