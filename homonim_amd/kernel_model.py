"""
Sliding-window kernel models on the MI355X: host-side mirror of the reference's operator interface for the hot path.

Same class / method names, argument meaning and error behaviour as homonim/kernel_model.py (``KernelModel`` :35-463,
``RefSpaceModel`` :466-503, ``SrcSpaceModel`` :506-535), but ``fit`` / ``apply`` hand the block to hand-written HIP
kernels through the C ABI of include/homonim_hk.h (ctypes, ``homonim_amd/_hk.py``).  Nothing here computes pixels on
the CPU and there is no fallback: without the built library or a gfx950 device the calls raise ``DeviceError``.

Differences a caller can observe (all documented in DESIGN.md):

* ``fit`` does NOT zero-fill / normalise the caller's ``src_ra`` / ``ref_ra`` in place (the reference does,
  kernel_model.py:246-247,292-295,320-321; its own wrappers always pass temporaries or copies).
* ``fit_apply`` is an extra, fused entry point for ``RasterFuse._process_block``'s fit->apply pair (fuse.py:305-307).
* gain-offset with ``r2_inpaint_thresh`` set: the kernel evaluates the r2 mask (kernel_model.py:363); when no valid
  pixel fails it the reference's GDAL ``fillnodata`` branch is the identity and results are identical; when some
  fail, the offsets are in-painted on the device by a restatement of GDAL's published fill algorithm (parity with
  GDAL itself unpinned, hk_inpaint.hip) and the gains of the failing pixels are recomputed as in :370-371.
* ``RefSpaceModel`` / ``SrcSpaceModel`` re-sample between same-CRS, axis-aligned grids on the device (every
  ``rasterio.enums.Resampling`` warp method: a restatement of GDAL's warp kernels, pinned against the real stack's
  published accuracy table for average / cubic_spline, otherwise unpinned); other CRSs and rotations raise
  ``NotImplementedError``, ``gauss`` is rejected as in rasterio.
"""
from typing import Dict, Optional, Tuple

import numpy as np

from homonim_amd import _hk, utils
from homonim_amd.enums import Model, Resampling
from homonim_amd.raster_array import RasterArray


class KernelModel:
    default_kernel_shape = (5, 5)
    default_model = Model.gain_blk_offset

    def __init__(self, model: Model = default_model, kernel_shape: Tuple[int, int] = default_kernel_shape,
                 find_r2: bool = False, **kwargs):
        """
        model: correction model; kernel_shape: (height, width) of the sliding window, odd; find_r2: add an R2 band to
        the parameters; kwargs: see ``create_config`` (unknown keys raise TypeError, as in the reference).
        """
        self._model = Model(model)
        self._kernel_shape = utils.validate_kernel_shape(kernel_shape, model=self._model)
        self._find_r2 = find_r2
        config = self.create_config(**kwargs)
        self._r2_inpaint_thresh: Optional[float] = config['r2_inpaint_thresh']
        self._mask_partial: bool = config['mask_partial']
        self._downsampling: Resampling = config['downsampling']
        self._upsampling: Resampling = config['upsampling']
        self._ctx = None  # GPU context, created on first use

    # -- configuration (kernel_model.py:83-136) -----------------------------------------------------------------------
    @property
    def model(self) -> Model:
        return self._model

    @property
    def kernel_shape(self) -> Tuple[int, int]:
        return tuple(self._kernel_shape)

    @property
    def find_r2(self) -> bool:
        return self._find_r2

    @staticmethod
    def create_config(r2_inpaint_thresh: Optional[float] = 0.25, mask_partial: bool = False,
                      downsampling: Resampling = Resampling.average,
                      upsampling: Resampling = Resampling.cubic_spline) -> Dict:
        """ The model configuration dict accepted by ``__init__`` and ``RasterFuse.process`` (defaults = reference). """
        return dict(r2_inpaint_thresh=r2_inpaint_thresh, mask_partial=mask_partial, downsampling=downsampling,
                    upsampling=upsampling)

    def _get_resampling(self, from_res: Tuple[float, float], to_res: Tuple[float, float]):
        """ kernel_model.py:138-140 """
        return self._downsampling if np.prod(np.abs(from_res)) <= np.prod(np.abs(to_res)) else self._upsampling

    # -- device plumbing ----------------------------------------------------------------------------------------------
    @property
    def context(self) -> '_hk.Context':
        if self._ctx is None:
            self._ctx = _hk.default_context()
        return self._ctx

    @context.setter
    def context(self, ctx: '_hk.Context'):
        self._ctx = ctx

    @property
    def _emit_r2(self) -> bool:
        # kernel_model.py:252 (gain models) / :325 (gain-offset)
        return bool(self._find_r2 or (self._model == Model.gain_offset and self._r2_inpaint_thresh is not None))

    def _desc(self, src_ra: RasterArray, ref_ra: RasterArray):
        thresh = self._r2_inpaint_thresh if self._model == Model.gain_offset else None
        return _hk.make_desc(self._model, self._kernel_shape, self._find_r2, thresh, src_ra.nodata, ref_ra.nodata)

    @staticmethod
    def _band(ra: RasterArray, what: str) -> np.ndarray:
        arr = ra.array
        if arr.ndim == 3:
            if arr.shape[0] != 1:
                raise ValueError(f'`{what}` must hold a single band')
            arr = arr[0]
        return arr

    def _param_profile(self, src_ra: RasterArray, count: int) -> Dict:
        profile = src_ra.profile.copy()
        profile.update(count=count, nodata=RasterArray.default_nodata, dtype=RasterArray.default_dtype)
        return profile

    # -- the hot path -------------------------------------------------------------------------------------------------
    def fit(self, src_ra: RasterArray, ref_ra: RasterArray) -> RasterArray:
        """
        Fit sliding kernel models to a source / reference block pair on the same grid (kernel_model.py:411-440).
        Returns the parameters: gains in band 0, offsets in band 1 and, when ``find_r2`` (or gain-offset with an
        ``r2_inpaint_thresh``), R2 in band 2; NaN where either input is nodata.
        """
        if (ref_ra.transform != src_ra.transform) or (ref_ra.shape != src_ra.shape):
            raise ValueError("'ref_ra' and 'src_ra' must have the same CRS, transform and shape")
        count = 3 if self._emit_r2 else 2
        params, _, _, n_fail = self.context.fit_apply(
            self._desc(src_ra, ref_ra), self._band(src_ra, 'src_ra'), self._band(ref_ra, 'ref_ra'), count,
            want_params=True, want_corr=False
        )
        return RasterArray.from_profile(params, self._param_profile(src_ra, count))

    def apply(self, src_ra: RasterArray, param_ra: RasterArray) -> RasterArray:
        """ Corrected block = gain * source + offset (kernel_model.py:442-463). """
        if (param_ra.transform != src_ra.transform) or (param_ra.shape != src_ra.shape):
            raise ValueError("'param_ra' and 'src_ra' must have the same CRS, transform and shape")
        corr = self.context.apply(self._band(src_ra, 'src_ra'), param_ra.array)
        return RasterArray.from_profile(corr, param_ra.profile)

    def fit_apply(self, src_ra: RasterArray, ref_ra: RasterArray, want_params: bool = False,
                  out_dtype: str = RasterArray.default_dtype,
                  out_nodata: Optional[float] = RasterArray.default_nodata) -> Tuple[RasterArray, Optional[RasterArray]]:
        """
        ``apply(src_ra, fit(src_ra.copy(), ref_ra))`` in ONE pass over the data (window sums, solve, R2 test and
        correction fused in a single kernel; each input byte is read once).  Returns (corr_ra, param_ra | None).

        ``out_dtype`` / ``out_nodata`` convert the corrected block on the device the way the reference converts it when
        writing (raster_array.py:353-387); integer ``src_ra`` / ``ref_ra`` arrays are converted to float32 on the device.
        """
        if (ref_ra.transform != src_ra.transform) or (ref_ra.shape != src_ra.shape):
            raise ValueError("'ref_ra' and 'src_ra' must have the same CRS, transform and shape")
        count = 3 if self._emit_r2 else 2
        params, corr, _, n_fail = self.context.fit_apply(
            self._desc(src_ra, ref_ra), self._band(src_ra, 'src_ra'), self._band(ref_ra, 'ref_ra'), count,
            want_params=want_params, want_corr=True, out_dtype=out_dtype, out_nodata=out_nodata
        )
        profile = self._param_profile(src_ra, count)
        corr_profile = dict(profile, nodata=out_nodata, dtype=str(out_dtype))
        corr_ra = RasterArray.from_profile(corr, corr_profile)
        return corr_ra, (RasterArray.from_profile(params, profile) if want_params else None)

    def fuses_into(self, src_ra: RasterArray, ref_ra: RasterArray) -> bool:
        """ True when ``fit_apply_into`` covers this model configuration for the pair (base-class fit + apply on a shared
        grid, nothing between them): the wrappers narrow it. """
        return (ref_ra.transform == src_ra.transform) and (ref_ra.shape == src_ra.shape)

    def fit_apply_into(self, src_ra: RasterArray, ref_ra: RasterArray, window, corr_dst: np.ndarray,
                       params_dst: Optional[np.ndarray] = None, out_nodata: Optional[float] = RasterArray.default_nodata) -> int:
        """
        ``fit_apply`` for the block loop of ``RasterFuse.process`` (homonim/fuse.py:295-319): the window
        ``(row0, col0, rows, cols)`` of the block -- its out-block, i.e. the block without the halo that
        raster_array.py:478-491 crops on writing -- goes straight into ``corr_dst`` / ``params_dst``, views of the
        caller's corrected / parameter rasters (their dtype is the output dtype).  One call = one upload of the read-block,
        one download of the out-block, one stream synchronisation; asynchronous when the rasters are page-locked.
        Returns the number of pixels that failed the r2 mask.
        """
        if not KernelModel.fuses_into(self, src_ra, ref_ra):
            raise ValueError("'ref_ra' and 'src_ra' must have the same CRS, transform and shape")
        _, n_fail = self.context.fit_apply_block(
            self._desc(src_ra, ref_ra), self._band(src_ra, 'src_ra'), self._band(ref_ra, 'ref_ra'), window, corr_dst,
            params_dst, out_nodata=out_nodata
        )
        return n_fail

    def block_norm(self, src_ra: RasterArray, ref_ra: RasterArray) -> np.ndarray:
        """ The [gain, offset] block normalisation of gain-blk-offset (kernel_model.py:216-229), float64[2]. """
        return self.context.block_norm(self._desc(src_ra, ref_ra), self._band(src_ra, 'src_ra'),
                                       self._band(ref_ra, 'ref_ra'))


def _same_grid(a: RasterArray, b: RasterArray) -> bool:
    return a.transform == b.transform and a.shape == b.shape and a.crs == b.crs


def _full_coverage_mask(model: KernelModel, in_mask_ra: RasterArray, param_ra: RasterArray) -> np.ndarray:
    """
    kernel_model.py:375-409: the mask of parameter-grid pixels fully covered by the input mask (re-projected with
    `average`, >= 1), having parameters, and whose (kh+2) x (kw+2) neighbourhood is all of that kind (erosion with a zero
    border).  Returns a bool array on the parameter grid; evaluated on the device.
    """
    cover_ra = in_mask_ra.reproject(**param_ra.proj_profile, nodata=None, resampling=Resampling.average,
                                    context=model.context)
    _, _, mask = model.context.partial_mask(cover_ra.array, None, param_ra.array[:2], model.kernel_shape, want_mask=True,
                                            coverage=True)
    return mask.astype(bool)


class RefSpaceModel(KernelModel):
    """
    Parameters estimated on the reference grid (kernel_model.py:466-503): the source block is re-sampled to the
    reference grid for ``fit`` and the parameters are re-sampled back to the source grid for ``apply``.  Re-sampling
    runs on the GPU for same-CRS, axis-aligned grids (hk_resample.hip: a restatement of GDAL's warp kernels).
    """

    def fit(self, src_ra: RasterArray, ref_ra: RasterArray) -> RasterArray:
        if not _same_grid(src_ra, ref_ra):  # identical grids: the reference's `average` re-sampling is the identity
            resampling = self._get_resampling(src_ra.res, ref_ra.res)
            src_ra = src_ra.reproject(**ref_ra.proj_profile, resampling=resampling, context=self.context)  # :480
        return KernelModel.fit(self, src_ra, ref_ra)

    def apply(self, src_ra: RasterArray, param_ra: RasterArray) -> RasterArray:
        if _same_grid(src_ra, param_ra):
            if self._mask_partial:
                # full-coverage mask of the source mask & parameters, eroded by (kh+2) x (kw+2), then apply (:493-503)
                _, corr, _ = self.context.partial_mask(
                    self._band(src_ra, 'src_ra'), src_ra.nodata, param_ra.array[:2], self._kernel_shape,
                    src=self._band(src_ra, 'src_ra'), want_corr=True
                )
                return RasterArray.from_profile(corr, param_ra.profile)
            # the reference keeps only gain & offset and re-masks them with the source mask (:487,:500); on a shared
            # grid the parameters are already nodata wherever the source is.
            return KernelModel.apply(self, src_ra, param_ra)

        _param_ra = RasterArray.from_profile(param_ra.array[:2], param_ra.profile)  # :487
        resampling = self._get_resampling(param_ra.res, src_ra.res)
        param_us_ra = _param_ra.reproject(**src_ra.proj_profile, resampling=resampling, context=self.context)  # :491
        if self._mask_partial:
            mask = _full_coverage_mask(self, src_ra.mask_ra, _param_ra)  # :495
            mask_ra = RasterArray(mask.astype('float32'), _param_ra.crs, _param_ra.transform, nodata=None)
            mask_us_ra = mask_ra.reproject(**src_ra.proj_profile, nodata=0, resampling=Resampling.nearest,
                                           context=self.context)  # :497
            param_us_ra.mask = mask_us_ra.array.astype('bool', copy=False)  # :498
        else:
            param_us_ra.mask = src_ra.mask  # :500
        return KernelModel.apply(self, src_ra, param_us_ra)


    def fuses_into(self, src_ra: RasterArray, ref_ra: RasterArray) -> bool:
        return _same_grid(src_ra, ref_ra) and not self._mask_partial

    def fit_apply(self, src_ra: RasterArray, ref_ra: RasterArray, want_params: bool = False,
                  out_dtype: str = RasterArray.default_dtype,
                  out_nodata: Optional[float] = RasterArray.default_nodata) -> Tuple[RasterArray, Optional[RasterArray]]:
        """ ``apply(src_ra, fit(src_ra, ref_ra))`` with everything between the two blocks kept in HBM
        (hk_refspace_fit_apply): one upload of source + reference, one download of the corrected block. """
        if _same_grid(src_ra, ref_ra) and not self._mask_partial:
            return KernelModel.fit_apply(self, src_ra, ref_ra, want_params, out_dtype, out_nodata)
        from homonim_amd.geo import grid_mapping
        if src_ra.crs != ref_ra.crs:
            raise NotImplementedError('re-projection between different CRSs is not built (GDAL warp)')
        down = self._get_resampling(src_ra.res, ref_ra.res)
        up = self._get_resampling(ref_ra.res, src_ra.res)
        count = 3 if self._emit_r2 else 2
        params, corr, _ = self.context.refspace_fit_apply(
            self._desc(src_ra, ref_ra), self._band(src_ra, 'src_ra'), self._band(ref_ra, 'ref_ra'),
            grid_mapping(src_ra.transform, ref_ra.transform), grid_mapping(ref_ra.transform, src_ra.transform), int(down),
            int(up), self._mask_partial, count, want_params, out_dtype=out_dtype, out_nodata=out_nodata
        )
        corr_profile = dict(self._param_profile(src_ra, 1), nodata=out_nodata, dtype=str(out_dtype))
        corr_ra = RasterArray.from_profile(corr, corr_profile)
        param_ra = RasterArray.from_profile(params, self._param_profile(ref_ra, count)) if want_params else None
        return corr_ra, param_ra


class SrcSpaceModel(KernelModel):
    """ Parameters estimated on the source grid (kernel_model.py:506-535): the reference block is re-sampled to it. """

    def fit(self, src_ra: RasterArray, ref_ra: RasterArray) -> RasterArray:
        ref_us_ra = ref_ra
        if not _same_grid(src_ra, ref_ra):
            resampling = self._get_resampling(ref_ra.res, src_ra.res)
            ref_us_ra = ref_ra.reproject(**src_ra.proj_profile, resampling=resampling, context=self.context)  # :520
        param_ra = KernelModel.fit(self, src_ra, ref_us_ra)
        if self._mask_partial:
            # full-coverage mask of the reference mask & gain/offset, eroded by (kh+2) x (kw+2), on all bands (:526-531)
            if _same_grid(src_ra, ref_ra):
                masked, _, _ = self.context.partial_mask(
                    self._band(ref_ra, 'ref_ra'), ref_ra.nodata, param_ra.array, self._kernel_shape, want_params=True
                )
                param_ra = RasterArray.from_profile(masked, param_ra.profile)
            else:
                param_ra.mask = _full_coverage_mask(self, ref_ra.mask_ra, param_ra)
        elif not _same_grid(src_ra, ref_ra):
            param_ra.mask = src_ra.mask  # :533 (a no-op on a shared grid)
        return param_ra

    def fuses_into(self, src_ra: RasterArray, ref_ra: RasterArray) -> bool:
        return _same_grid(src_ra, ref_ra) and not self._mask_partial

    def fit_apply(self, src_ra: RasterArray, ref_ra: RasterArray, want_params: bool = False,
                  out_dtype: str = RasterArray.default_dtype,
                  out_nodata: Optional[float] = RasterArray.default_nodata) -> Tuple[RasterArray, Optional[RasterArray]]:
        """ One fused pass on a shared grid; with ``mask_partial`` (or across grids) the reference's own sequence
        ``apply(src_ra, fit(src_ra, ref_ra))`` -- SrcSpaceModel.fit masks the parameters with the eroded full-coverage
        mask (kernel_model.py:526-531), which the fused kernel does not know about. """
        if self.fuses_into(src_ra, ref_ra):
            return KernelModel.fit_apply(self, src_ra, ref_ra, want_params, out_dtype, out_nodata)
        param_ra = self.fit(src_ra.copy(), ref_ra)
        corr_ra = self.apply(src_ra, param_ra)
        if np.dtype(out_dtype) != np.float32 or not (out_nodata is None or (isinstance(out_nodata, float) and np.isnan(out_nodata))):
            from homonim_amd.fuse import convert_dtype
            arr = convert_dtype(corr_ra.array, str(np.dtype(out_dtype)), out_nodata)
            corr_ra = RasterArray.from_profile(arr, dict(corr_ra.profile, nodata=out_nodata, dtype=str(np.dtype(out_dtype))))
        return corr_ra, (param_ra if want_params else None)
