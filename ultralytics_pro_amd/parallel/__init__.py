from .dp import (BucketedAllReduce, allreduce_gradients_, broadcast_, broadcast_flag, reduce_mean_, flat_parameter_layout, gradient_bucket_table, gather_detections, gather_ragged, init_distributed, max_over_ranks,  # noqa: F401
                 shard_first_image)
