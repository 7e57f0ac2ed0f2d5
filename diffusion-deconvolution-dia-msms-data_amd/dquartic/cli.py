"""``dquartic`` command line -- same commands and options as the reference's ``dquartic/cli.py`` (:26-188):
``dquartic train CONFIG [--parquet_directory --ms2-data-path --ms1-data-path --batch-size --checkpoint-path --use-wandb
--threads]`` and ``dquartic generate-config PATH``.  ``generate-train-data`` (offline sqMass ETL) is outside the hot path
and reports so.  Under ``torch.distributed.run`` (WORLD_SIZE > 1) training is data-parallel: one process per GPU, the
dataset is sharded by rank and the flat gradient is all-reduced over RCCL."""
import ast
import os

import click
import torch
from torch.utils.data import DataLoader

from .model.model import DDIMDiffusionModel
from .model.unet1d import UNet1d
from .utils.config_loader import generate_train_config, load_train_config


class PythonLiteralOption(click.Option):
    def type_cast_value(self, ctx, value):
        if not isinstance(value, str):
            return value
        try:
            return ast.literal_eval(value)
        except Exception:
            raise click.BadParameter(value)


@click.group(chain=True)
@click.version_option(package_name=None, version="0.1.0")
def cli():
    """Diffusion Deconvolution of DIA-MS/MS Data (D^4) -- MI355X build"""


@cli.command()
@click.argument("config-path", type=click.Path(exists=True), required=True)
@click.option("--parquet_directory", default=None, help="Directory of parquet slices (overrides the config)")
@click.option("--ms2-data-path", default=None, help="Path to MS2 .npy data (overrides the config)")
@click.option("--ms1-data-path", default=None, help="Path to MS1 .npy data (overrides the config)")
@click.option("--batch-size", default=None, help="Batch size (overrides the config)")
@click.option("--checkpoint-path", default=None, help="Where to save the best model (overrides the config)")
@click.option("--use-wandb", default=None, cls=PythonLiteralOption, help="Use wandb for logging (overrides the config)")
@click.option("--threads", default=None, help="Data-loading worker processes (overrides the config)")
@click.option("--resident-dataset", is_flag=True, default=False,
              help="Keep the whole dataset in HBM and form batches on the GPU (dq_pair_batch) instead of a DataLoader")
def train(config_path, parquet_directory, ms2_data_path, ms1_data_path, batch_size, checkpoint_path, use_wandb, threads,
          resident_dataset):
    """Train a DDIM model on DIA-MS windows."""
    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise click.ClickException("no GPU visible: this build runs the hot path on MI355X only (no CPU fallback)")
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local))
    if rank == 0:
        click.echo("--" * 30)
        for i in range(torch.cuda.device_count()):
            click.echo(f"GPU {i}: {torch.cuda.get_device_name(i)}  {torch.cuda.get_device_properties(i).total_memory / 2**20:.0f} MB")
        click.echo("--" * 30)
        click.echo(f"Info: Loading config from {config_path}")
    config = load_train_config(config_path, parquet_directory=parquet_directory, ms2_data_path=ms2_data_path,
                               ms1_data_path=ms1_data_path, batch_size=batch_size, checkpoint_path=checkpoint_path,
                               use_wandb=use_wandb, threads=threads)
    m = config["model"]
    syn = config["data"].get("synthetic")
    if syn:
        from .utils.synthetic import SyntheticDIAMSDataset

        dataset = SyntheticDIAMSDataset(n_windows=int(syn.get("n_windows", 32)), RT=int(syn.get("RT", 400)),
                                        MZ=int(syn.get("MZ", m["UNet1d"]["downsample_dim"])), normalize=config["data"]["normalize"],
                                        rank=rank, world=world)
    else:
        from .utils.data_loader import DIAMSDataset

        dataset = DIAMSDataset(config["data"]["parquet_directory"], config["data"]["ms2_data_path"], config["data"]["ms1_data_path"],
                               normalize=config["data"]["normalize"])
    per_rank = max(1, int(m["batch_size"]) // world)
    device = torch.device("cuda", local)
    if resident_dataset and not syn:
        from .utils.data_loader import ResidentPairLoader

        loader = ResidentPairLoader(dataset, per_rank, device=device, drop_last=len(dataset) // world > per_rank, rank=rank, world_size=world)
    elif world > 1 and not syn:
        # data-parallel over a file-backed dataset: rank r draws the indices i with i % world == r of a per-epoch permutation
        # (SURVEY 8e).  (DIAMSDataset ignores the index and returns a random pair, data_loader.py:60-69, so this fixes the number
        # of batches per rank and epoch rather than the windows a rank sees; the synthetic dataset shards its window pool itself.)
        from torch.utils.data.distributed import DistributedSampler

        sampler = DistributedSampler(dataset, num_replicas=world, rank=rank, shuffle=True, drop_last=False)
        loader = DataLoader(dataset, batch_size=per_rank, sampler=sampler, num_workers=int(config["threads"]),
                            drop_last=len(dataset) // world > per_rank)
    else:
        loader = DataLoader(dataset, batch_size=per_rank, shuffle=True, num_workers=int(config["threads"]), drop_last=len(dataset) > per_rank)
    if m["use_model"] == "UNet1d":
        u = m["UNet1d"]
        net = UNet1d(dim=u["dim"], channels=u["channels"], dim_mults=tuple(u["dim_mults"]), conditional=u["conditional"],
                     init_cond_channels=u["init_cond_channels"], attn_cond_channels=u["attn_cond_channels"],
                     tfer_dim_mult=u["tfer_dim_mult"], downsample_dim=u["downsample_dim"], simple=u["simple"]).to(device)
    elif m["use_model"] == "CustomTransformer":  # reference cli.py:102-109; served through the 4-argument adapter
        from .model.building_blocks import CustomTransformer, DDIMTransformerAdapter

        c = m["CustomTransformer"]
        net = DDIMTransformerAdapter(CustomTransformer(input_dim=c["input_dim"], hidden_dim=c["hidden_dim"], num_heads=c["num_heads"],
                                                       num_layers=c["num_layers"])).to(device)
    else:
        raise click.ClickException(f"Invalid model class: {m['use_model']}")  # reference cli.py:111 (ValueError there)
    dm = DDIMDiffusionModel(model_class=net, num_timesteps=m["num_timesteps"], beta_schedule_type=m["beta_schedule_type"],
                            pred_type=m["pred_type"], auto_normalize=m["auto_normalize"], ms1_loss_weight=m["ms1_loss_weight"],
                            device=device)
    wb = None
    if config["wandb"]["use_wandb"] and rank == 0:
        try:
            import wandb as wb

            w = config["wandb"]
            wb.init(project=w["wandb_project"], name=w["wandb_name"], id=w["wandb_id"], resume=w["wandb_resume"],
                    config={"architecture": w["wandb_architecture"], "dataset": w["wandb_dataset"], **m}, mode=w["wandb_mode"])
        except ImportError:
            click.echo("wandb is not installed; continuing without it")
            wb = None
    dm.train(loader, m["batch_size"], m["num_epochs"], m["warmup_epochs"], m["learning_rate"], wb is not None, m["checkpoint_path"])
    if wb is not None:
        wb.finish()
    if world > 1:
        torch.distributed.destroy_process_group()


@cli.command()
@click.argument("config-path", type=click.Path(), required=True)
def generate_config(config_path):
    """Write the default training configuration."""
    click.echo(f"Info: Generating config at {config_path}")
    generate_train_config(config_path)


@cli.command()
@click.argument("input-file", type=click.Path(), required=True)
@click.argument("output-file", type=click.Path(), required=True)
def generate_train_data(input_file, output_file):
    """(not built) sqMass -> parquet slice ETL."""
    raise click.ClickException("generate-train-data is the reference's offline ETL on instrument files; it is outside the "
                                "accelerated hot path (SURVEY section 2) -- use the reference tool to produce parquet/npy slices")


if __name__ == "__main__":
    cli()
