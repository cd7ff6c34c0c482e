"""Data side of the mixed-resolution training configuration (BASELINE config 4): aspect buckets, the bucketed
batch schedule, its distributed sampler and the batch dictionary the engine consumes.  File scanning, PIL decoding and the
Lightning DataModule around them stay the reference's (`neurosis.dataset.imagefolder`)."""
from .aspect import AspectBucket, AspectBucketList, AspectDistributedSampler, SDXLBucketList, bucket_batch_schedule, collate_bucket_batch

__all__ = ["AspectBucket", "AspectBucketList", "AspectDistributedSampler", "SDXLBucketList", "bucket_batch_schedule", "collate_bucket_batch"]
