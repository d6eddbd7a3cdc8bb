from .bucket import BucketBatchSampler
