from .dp import gather_detections, init_distributed, max_over_ranks, shard_first_image  # noqa: F401
