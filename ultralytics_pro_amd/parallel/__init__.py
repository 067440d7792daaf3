from .dp import (allreduce_gradients_, gather_detections, init_distributed, max_over_ranks,  # noqa: F401
                 shard_first_image)
