"""
``RasterCompare``: the accuracy statistics the reference reports for a source / corrected image against its reference
(homonim/compare.py) -- the acceptance metric of the hot path (r2 up, RMSE and rRMSE down after correction,
tests/integration.py:79-83 of the reference).

Same surface as the reference class on in-memory rasters (``RasterFuse`` supplies the pair handling and the block
partition): ``schema``, ``create_config``, ``process(**config) -> {band: {r2, rmse, rrmse, n}, ..., 'Mean': {...}}``,
``stats_table``.  Per block the source (or the reference) is re-sampled onto the processing grid on the GPU
(compare.py:236-241) and the seven masked sums of compare.py:243-255 are reduced on the GPU (``hk_compare_sums``); the
statistics follow from the accumulated sums exactly as in compare.py:142-186.

Deviation: the block sums come back as float64 (exact sums of the float32 per-pixel terms) where numpy returns its
float32 pairwise sums; the statistics agree with the reference's to its own summation error (~1e-6 relative).
"""
import math
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, List, Optional, Sequence

import numpy as np

from homonim_amd import _hk, utils
from homonim_amd.enums import ProcCrs, Resampling
from homonim_amd.fuse import RasterFuse
from homonim_amd.geo import Affine
from homonim_amd.raster_array import RasterArray

_SUM_KEYS = ('src_sum', 'ref_sum', 'src2_sum', 'ref2_sum', 'src_ref_sum', 'res2_sum', 'mask_sum')


class RasterCompare(RasterFuse):
    """ Compare source and reference rasters (bands first, float32; see ``RasterFuse`` for the constructor). """

    schema = dict(
        r2=dict(abbrev='r\N{SUPERSCRIPT TWO}', description='Pearson\'s correlation coefficient squared'),
        rmse=dict(abbrev='RMSE', description='Root Mean Square Error'),
        rrmse=dict(abbrev='rRMSE', description='Relative RMSE (RMSE/mean(ref))'),
        n=dict(abbrev='N', description='Number of pixels'),
    )

    def __init__(self, *args, band_names: Optional[Sequence[str]] = None, **kwargs):
        RasterFuse.__init__(self, *args, **kwargs)
        self._band_names = list(band_names) if band_names is not None else None

    @staticmethod
    def schema_table() -> str:
        from tabulate import tabulate
        headers = {key: key.upper() for key in list(RasterCompare.schema.values())[0].keys()}
        return tabulate(RasterCompare.schema.values(), headers=headers, tablefmt='simple')

    @staticmethod
    def create_config(threads: int = 0, max_block_mem: float = 512, downsampling: Resampling = Resampling.average,
                      upsampling: Resampling = Resampling.cubic_spline) -> Dict:
        """ compare.py:99-131 """
        return dict(threads=utils.validate_threads(threads), max_block_mem=max_block_mem, downsampling=downsampling,
                    upsampling=upsampling)

    def _get_resampling(self, from_res, to_res, **kwargs) -> Resampling:
        """ compare.py:133-139: down-sampling method when going to the coarser grid, else the up-sampling one """
        config = self.create_config(**kwargs)
        return config['downsampling'] if np.prod(np.abs(from_res)) <= np.prod(np.abs(to_res)) else config['upsampling']

    @staticmethod
    def _band_stats(src_sum=0., ref_sum=0., src2_sum=0., ref2_sum=0., src_ref_sum=0., res2_sum=0., mask_sum=0.) -> Dict:
        """ compare.py:145-163 (Pearson r2 by the sample formula, RMSE, RMSE / mean(ref)); empty bands give NaN. """
        with np.errstate(divide='ignore', invalid='ignore'):
            n = np.float64(mask_sum)
            src_mean, ref_mean = np.float64(src_sum) / n, np.float64(ref_sum) / n
            pcc_num = src_ref_sum - (n * src_mean * ref_mean)
            pcc_den = np.sqrt(src2_sum - (n * (src_mean ** 2))) * np.sqrt(ref2_sum - (n * (ref_mean ** 2)))
            pcc = pcc_num / pcc_den
            rmse = np.sqrt(np.float64(res2_sum) / n)
            rrmse = rmse / ref_mean
        return dict(r2=float(pcc ** 2), rmse=float(rmse), rrmse=float(rrmse), n=int(mask_sum))

    def _get_image_stats(self, image_sums: List[Dict]) -> Dict[str, Dict]:
        """ compare.py:142-186: per-band statistics keyed by band name, plus their mean under 'Mean' """
        image_stats, sum_over_bands = {}, {}
        for band_i, band_sums in enumerate(image_sums):
            band_stats = self._band_stats(**band_sums)
            name = self._band_names[band_i] if self._band_names else f'Ref. band {band_i + 1}'
            image_stats[name] = band_stats
            sum_over_bands = {k: sum_over_bands.get(k, 0) + v for k, v in band_stats.items()}
        image_stats['Mean'] = {k: int(v / len(image_sums)) if isinstance(v, int) else (v / len(image_sums))
                               for k, v in sum_over_bands.items()}
        return image_stats

    @staticmethod
    def stats_table(stats_dict: Dict[str, Dict], key_header: str = 'band') -> str:
        """ compare.py:188-210 """
        from tabulate import tabulate
        stats_list = [dict(**{key_header: key}, **val) for key, val in stats_dict.items()]
        headers = {k: RasterCompare.schema[k]['abbrev'] if k in RasterCompare.schema else str.capitalize(k)
                   for k in list(stats_list[0].keys())}
        return tabulate(stats_list, headers=headers, floatfmt='.3f', stralign='right', tablefmt='simple')

    def _block_sums(self, bp, ctx, config) -> Dict:
        """ compare.py:232-256 for one block """
        if self._same_grid:
            src_ra, ref_ra = self._read(bp)
        else:
            s_arr, s_nd = self._read_boundless(self._src[bp.band_i], self._src_nodata, bp.src_in_block)
            r_arr, r_nd = self._read_boundless(self._ref[bp.band_i], self._ref_nodata, bp.ref_in_block)
            src_tf = self._transform * Affine.translation(bp.src_in_block.col_off, bp.src_in_block.row_off)
            ref_tf = self._ref_transform * Affine.translation(bp.ref_in_block.col_off, bp.ref_in_block.row_off)
            src_ra = RasterArray(np.ascontiguousarray(s_arr, dtype=np.float32), self._crs, src_tf, nodata=s_nd)
            ref_ra = RasterArray(np.ascontiguousarray(r_arr, dtype=np.float32), self._crs, ref_tf, nodata=r_nd)
            if self._proc_crs == ProcCrs.ref:
                resampling = self._get_resampling(src_ra.res, ref_ra.res, **config)
                src_ra = src_ra.reproject(**ref_ra.proj_profile, resampling=resampling, context=ctx)
            else:
                resampling = self._get_resampling(ref_ra.res, src_ra.res, **config)
                ref_ra = ref_ra.reproject(**src_ra.proj_profile, resampling=resampling, context=ctx)
        sums = ctx.compare_sums(src_ra.array, src_ra.nodata, ref_ra.array, ref_ra.nodata)
        return dict(zip(_SUM_KEYS, (float(v) for v in sums)))

    def process(self, device_config: Optional[Dict] = None, **kwargs) -> Dict[str, Dict]:
        """ Compare the rasters (compare.py:212-278).  ``kwargs``: see ``create_config``. """
        if self._closed:
            from homonim_amd.errors import IoError
            raise IoError('The raster pair has been closed')
        config = self.create_config(**kwargs)
        device_config = RasterFuse.create_device_config(**(device_config or {}))
        import os
        devices = device_config['devices']
        if devices is None:
            devices = [int(os.environ.get('HOMONIM_AMD_DEVICE', os.environ.get('LOCAL_RANK', '0')))]
        contexts = [_hk.get_context(dev, device_config['streams']) for dev in devices]
        blocks = list(self.block_pairs(max_block_mem=config['max_block_mem'] or math.inf))
        image_sums = [dict.fromkeys(_SUM_KEYS, 0.) for _ in range(self._src.shape[0])]

        def accumulate(bp, sums):
            acc = image_sums[bp.band_i]
            for k, v in sums.items():
                acc[k] += v

        if config['threads'] == 1 and len(contexts) == 1:
            for bp in blocks:
                accumulate(bp, self._block_sums(bp, contexts[0], config))
        else:
            with ThreadPoolExecutor(max_workers=max(config['threads'], len(contexts))) as ex:
                futures = [(bp, ex.submit(self._block_sums, bp, contexts[i % len(contexts)], config))
                           for i, bp in enumerate(blocks)]
                for bp, f in futures:  # block order: the accumulation is reproducible run to run
                    accumulate(bp, f.result())
        return self._get_image_stats(image_sums)
