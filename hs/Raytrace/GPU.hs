{-# LANGUAGE ForeignFunctionInterface #-}
-- | Haskell side of libraytrace_gpu.so: the reference's scene API (src/Main.hs) with `Shape` and `Light` as recorded
-- constructor calls instead of closures, and `parallelTrace` as ONE foreign call.
--
-- WRITTEN BLIND: neither this image nor the GPU box has GHC (probed: ghc, cabal, stack, runghc absent), so this module
-- has never been compiled.  It is the binding a maintainer of the reference would add; the same C entry points are
-- exercised through ctypes (raytrace_amd/_lib.py) and C++ (include/raytrace.hpp, examples/main.cpp) by the test suite.
-- See INTEGRATION.md.
--
-- The module keeps the reference's names and types wherever a closure did not force a change, so that the scene code of
-- src/Main.hs pastes in unchanged: `planes`, `axes`, `cubes`, `spheres`, `stacked_cubes`, `triangle_example`
-- (Main.hs:639-723), `spec_test`, `intersection`, `tree` (776-840), and `main` itself (569-621):
--
-- > main =
-- >   let trace' = parallelTrace w h camera (trace 4 lights world)      -- Main.hs:571, unchanged
-- >       (world,lights) = linearInterpolation f (Just fnorm) (-1.0,-1.0) (1.0,1.0) 0.2 (MkV3D 0.01 (-7) (-25)) 10
-- >       camera = fixedCamera w h
-- >       ...
-- >   in saveBmpImage "trace.bmp" $ ImageRGB8 $ generateImage trace' w h  -- Main.hs:621, unchanged
--
-- What differs from the reference, and why:
--   * `Shape` is a tree of constructor calls (the closure tree as data); `intersect`/`boundingBox` are gone.
--   * `Light` is a list of sources; `pointLight world ...` keeps its first argument (Main.hs:467): when it is the value
--     that is traced (every scene of the reference) shadow rays meet the traced world, any other Shape is handed to the
--     library as a shadow world of that light.
--   * `trace n lights world` is data (`Tracer`), not a function of a ray; `Camera` is data.
--   * `parallelTrace` renders the whole frame on the GPU the first time a pixel is asked for (Main.hs:629-635 does the
--     same with `computeP`), then serves lookups from the frame.
module Raytrace.GPU
  ( -- * Vector3D (Main.hs:20-86)
    Vector3D(..), minV, maxV, (*@), (*#), norm, normalize, scalar
    -- * Materials and colours (Main.hs:176-185, 322-355)
  , Material(..), Color(..)
    -- * Shapes (Main.hs:189-415, 725-774)
  , Shape, rectangle, triangle, sphere, cube, colorcube, cuboid, colorcuboid, cylinder
    -- * Lights (Main.hs:418-492)
  , Light, pointLight, ambient
    -- * Camera, tracer, the per-pixel loop (Main.hs:494-557, 623-637)
  , Camera, fixedCamera, Tracer, trace, parallelTrace, parallelTraceMulti
    -- * Height fields (Main.hs:937-988)
  , linearInterpolation
    -- * Lower level
  , renderBand, initDevice
  ) where

import Codec.Picture (PixelRGB8 (..))
import Control.Exception (bracket)
import Control.Monad (forM, forM_, void, when)
import Data.Int (Int32)
import Data.IORef
import Data.Word (Word64, Word8)
import Foreign
import Foreign.C.String (CString, peekCString)
import Foreign.C.Types
import System.IO.Unsafe (unsafePerformIO)
import System.Mem.StableName (makeStableName)

-------------------------------------------------------------------------------
-- Vector3D — Main.hs:20-86, verbatim semantics

data Vector3D = MkV3D {-# UNPACK #-} !Double {-# UNPACK #-} !Double {-# UNPACK #-} !Double
  deriving (Eq, Show)

instance Num Vector3D where                                          -- Main.hs:31-42
  MkV3D x1 y1 z1 + MkV3D x2 y2 z2 = MkV3D (x1+x2) (y1+y2) (z1+z2)
  MkV3D x1 y1 z1 * MkV3D x2 y2 z2 = MkV3D (x1*x2) (y1*y2) (z1*z2)
  MkV3D x1 y1 z1 - MkV3D x2 y2 z2 = MkV3D (x1-x2) (y1-y2) (z1-z2)
  fromInteger i = MkV3D d d d where d = fromInteger i
  abs = undefined
  signum (MkV3D x y z) = MkV3D (signum x) (signum y) (signum z)

instance Fractional Vector3D where                                   -- Main.hs:44-48
  MkV3D x1 y1 z1 / MkV3D x2 y2 z2 = MkV3D (x1/x2) (y1/y2) (z1/z2)
  recip (MkV3D x y z) = MkV3D (recip x) (recip y) (recip z)
  fromRational r = MkV3D r' r' r' where r' = fromRational r

minV, maxV :: Vector3D -> Vector3D -> Vector3D                       -- Main.hs:51-58
minV (MkV3D x1 y1 z1) (MkV3D x2 y2 z2) = MkV3D (min x1 x2) (min y1 y2) (min z1 z2)
maxV (MkV3D x1 y1 z1) (MkV3D x2 y2 z2) = MkV3D (max x1 x2) (max y1 y2) (max z1 z2)

-- (no fixity declarations, as in the reference: infixl 9, tighter than *)
(*@) :: Vector3D -> Vector3D -> Double                               -- Main.hs:62-64
MkV3D x1 y1 z1 *@ MkV3D x2 y2 z2 = x1*x2 + y1*y2 + z1*z2

(*#) :: Vector3D -> Vector3D -> Vector3D                             -- Main.hs:67-70
MkV3D x1 y1 z1 *# MkV3D x2 y2 z2 = MkV3D (y1*z2-z1*y2) (z1*x2-x1*z2) (x1*y2-x2*y1)

norm :: Vector3D -> Double                                           -- Main.hs:73-74
norm v = sqrt (v *@ v)

normalize :: Vector3D -> Vector3D                                    -- Main.hs:78-80
normalize v@(MkV3D x y z) = MkV3D (x / n) (y/n) (z/n) where n = norm v

scalar :: Double -> Vector3D -> Vector3D                             -- Main.hs:84-85
scalar d (MkV3D x y z) = MkV3D (d*x) (d*y) (d*z)

-------------------------------------------------------------------------------
-- Materials and colours — Main.hs:176-185, 322-355

data Material = MkMaterial
  { mat_diffuse      :: !PixelRGB8
  , mat_specular     :: !PixelRGB8
  , mat_specularity  :: !Double
  , mat_reflectivity :: !Double
  }

class Color c where
  black,white,red,green,blue,magenta,cyan,yellow,orange,orchid,aquamarine :: c

instance Color PixelRGB8 where
  black   = PixelRGB8 0 0 0
  white   = PixelRGB8 255 255 255
  red     = PixelRGB8 255 0   0
  green   = PixelRGB8 0   255 0
  blue    = PixelRGB8 0   0   255
  magenta = PixelRGB8 255 0   255
  cyan    = PixelRGB8 0   255 255
  yellow  = PixelRGB8 255 255 0
  orange  = PixelRGB8 255 134 0
  orchid  = PixelRGB8 153 50  204
  aquamarine = PixelRGB8 69 139 116

instance Color Material where
  black   = MkMaterial black black 1.0 0.0
  white   = MkMaterial white white 1.0 0.0
  red     = MkMaterial red   red   1.0 0.0
  green   = MkMaterial green green 1.0 0.0
  blue    = MkMaterial blue  blue  1.0 0.0
  magenta = MkMaterial magenta magenta 1.0 0.0
  cyan    = MkMaterial cyan    cyan    1.0 0.0
  yellow  = MkMaterial yellow  yellow  1.0 0.0
  orange  = MkMaterial orange  orange  1.0 0.0
  orchid  = MkMaterial orchid  orchid  1.0 0.0
  aquamarine = MkMaterial aquamarine aquamarine 1.0 0.0

-------------------------------------------------------------------------------
-- Shapes: the closure tree of Main.hs:189-217 as data.  `<>` and `mconcat` keep the tree shape, because every mappend
-- node of the reference guards its subtree with a bounding-box test (Main.hs:205-216) and the library reproduces it.

data Shape
  = Rect Material Vector3D Vector3D Vector3D                       -- Main.hs:226
  | Tri Material Vector3D Vector3D Vector3D                        -- Main.hs:363
  | Sph Material Vector3D Double                                   -- Main.hs:392
  | Cuboid [Material] Vector3D Double Double Double                -- Main.hs:301
  | Cyl Material Material Material Vector3D Int Double Double      -- Main.hs:725
  | Append Shape Shape                                             -- Main.hs:205
  | Concat [Shape]                                                 -- mconcat = foldr mappend mempty
  | Normals (Double -> Double -> Vector3D) Double Vector3D Shape   -- Main.hs:956-965: fnorm', scaleInv, offset

instance Semigroup Shape where
  (<>) = Append
instance Monoid Shape where
  mempty  = Concat []
  mappend = Append
  mconcat = Concat

rectangle :: Material -> Vector3D -> Vector3D -> Vector3D -> Shape
rectangle = Rect
triangle :: Material -> Vector3D -> Vector3D -> Vector3D -> Shape
triangle = Tri
sphere :: Material -> Vector3D -> Double -> Shape
sphere = Sph
colorcuboid :: [Material] -> Vector3D -> Double -> Double -> Double -> Shape
colorcuboid [] _ _ _ _ = error "colorcuboid: list of materials must not be empty."   -- Main.hs:302
colorcuboid ms p l h d = Cuboid ms p l h d
cuboid :: Material -> Vector3D -> Double -> Double -> Double -> Shape
cuboid material point l h d = colorcuboid [material] point l h d
colorcube :: [Material] -> Vector3D -> Double -> Shape
colorcube materials point l = colorcuboid materials point l l l
cube :: Material -> Vector3D -> Double -> Shape
cube material point l = colorcube [material] point l
cylinder :: Material -> Material -> Material -> Vector3D -> Int -> Double -> Double -> Shape
cylinder = Cyl

-------------------------------------------------------------------------------
-- Lights — Main.hs:418-492

data LightSrc = Point Shape Double Double Vector3D | Ambient Double   -- a point light closes over the world its shadow rays meet
newtype Light = MkLight [LightSrc]
instance Semigroup Light where
  MkLight a <> MkLight b = MkLight (a ++ b)
instance Monoid Light where
  mempty = MkLight []

-- | Main.hs:467.  The first argument is the world the light's shadow rays are cast against.  When it is the very value
-- that is traced (every scene of the reference), the library tests them against the traced world; any other Shape is
-- handed over as a shadow world of its own (rt_begin_shadow_world ... rt_add_point_light_world).
pointLight :: Shape -> Double -> Double -> Vector3D -> Light
pointLight world kd ks lpos = MkLight [Point world kd ks lpos]
ambient :: Double -> Light                                          -- Main.hs:490
ambient f = MkLight [Ambient f]

-------------------------------------------------------------------------------
-- Camera and tracer as data — Main.hs:494-557

data Camera = MkCamera Int Int
fixedCamera :: Int -> Int -> Camera                                  -- Main.hs:506
fixedCamera = MkCamera

data Tracer = MkTracer Int Light Shape
trace :: Int -> Light -> Shape -> Tracer                             -- Main.hs:541 (the ray argument is the GPU's business)
trace = MkTracer

-------------------------------------------------------------------------------
-- linearInterpolation — Main.hs:937-988, with the reference's own arithmetic

linearInterpolation
  :: (Double -> Double -> Double) -> Maybe (Double -> Double -> Vector3D)
  -> (Double,Double) -> (Double,Double) -> Double -> Vector3D -> Double -> (Shape,Light)
linearInterpolation f fnorm (x1,y1) (x2,y2) step origin scale =
  let grid = [(x,y) | x <- steps x1 x2, y <- steps y1 y2 ] where
        steps a b = takeWhile (< b) (iterate (+ step) a)
      triangles =
        [ normals $ mconcat [
              triangle mat (f' x y') (f' x' y') (f' x y),
              triangle mat (f' x y) (f' x' y') (f' x' y) ]
          | (x,y) <- grid, let x' = x + step, let y' = y + step ]
      normals shape = case fnorm of
        Nothing -> shape
        Just fnorm' -> Normals fnorm' scaleInv offset shape          -- evaluated on the host for the GPU's hit points
      shape = mconcat [
        mconcat triangles,
        (rectangle aquamarine origin (MkV3D (1.5 * scale * w) 0 0) (MkV3D 0 0 (-1.5 * scale * h)))]
      light = mconcat [
        pointLight shape 0.3 0.6 (origin + MkV3D (scale * w/4) (scale * 10) (scale * w/4)),
        ambient 0.3]
      mat = white{mat_specularity = 1}
      f' a b = offset + scalar scale (MkV3D a (f a b) b)
      offset = origin - scalar scale (MkV3D x 0 y)
      x = (x2 + x1) / 2
      y = (y2 + y1) / 2
      w = (x2 - x1)
      h = (y2 - y1)
      scaleInv = recip scale
  in (shape,light)

-------------------------------------------------------------------------------
-- C ABI (include/raytrace_gpu.h)

data RtScene
type ScenePtr = Ptr RtScene
-- void (*rt_normal_fn)(int n, const double *xy, double *normals, void *user)
type NormalFn = CInt -> Ptr CDouble -> Ptr CDouble -> Ptr () -> IO ()

foreign import ccall unsafe "rt_init"            c_init            :: CInt -> IO CInt
foreign import ccall unsafe "rt_last_error"      c_last_error      :: IO CString
foreign import ccall unsafe "rt_scene_create"    c_scene_create    :: Ptr ScenePtr -> IO CInt
foreign import ccall unsafe "rt_scene_destroy"   c_scene_destroy   :: ScenePtr -> IO ()
foreign import ccall unsafe "rt_scene_num_leaves" c_num_leaves     :: ScenePtr -> IO CInt
foreign import ccall unsafe "rt_begin_mconcat"   c_begin_mconcat   :: ScenePtr -> IO CInt
foreign import ccall unsafe "rt_end_mconcat"     c_end_mconcat     :: ScenePtr -> IO CInt
foreign import ccall unsafe "rt_begin_mappend"   c_begin_mappend   :: ScenePtr -> IO CInt
foreign import ccall unsafe "rt_end_mappend"     c_end_mappend     :: ScenePtr -> IO CInt
foreign import ccall unsafe "rt_add_rectangle"   c_add_rectangle   :: ScenePtr -> Ptr Word8 -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> IO CInt
foreign import ccall unsafe "rt_add_triangle"    c_add_triangle    :: ScenePtr -> Ptr Word8 -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> IO CInt
foreign import ccall unsafe "rt_add_sphere"      c_add_sphere      :: ScenePtr -> Ptr Word8 -> Ptr CDouble -> CDouble -> IO CInt
foreign import ccall unsafe "rt_add_cuboid"      c_add_cuboid      :: ScenePtr -> Ptr Word8 -> CInt -> Ptr CDouble -> CDouble -> CDouble -> CDouble -> IO CInt
foreign import ccall unsafe "rt_add_cylinder"    c_add_cylinder    :: ScenePtr -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> Ptr CDouble -> CInt -> CDouble -> CDouble -> IO CInt
foreign import ccall unsafe "rt_set_normal_callback" c_set_normal_callback :: ScenePtr -> CInt -> CInt -> Ptr CDouble -> FunPtr NormalFn -> Ptr () -> IO CInt
foreign import ccall "wrapper"                   mkNormalFn        :: NormalFn -> IO (FunPtr NormalFn)
foreign import ccall unsafe "rt_add_point_light" c_add_point_light :: ScenePtr -> CDouble -> CDouble -> Ptr CDouble -> IO CInt
foreign import ccall unsafe "rt_add_ambient"     c_add_ambient     :: ScenePtr -> CDouble -> IO CInt
foreign import ccall unsafe "rt_begin_shadow_world" c_begin_shadow_world :: ScenePtr -> IO CInt
foreign import ccall unsafe "rt_end_shadow_world"   c_end_shadow_world   :: ScenePtr -> IO CInt
foreign import ccall unsafe "rt_add_point_light_world" c_add_point_light_world :: ScenePtr -> CInt -> CDouble -> CDouble -> Ptr CDouble -> IO CInt
foreign import ccall unsafe "rt_scene_set_option" c_scene_set_option :: ScenePtr -> CInt -> CInt -> IO CInt
foreign import ccall unsafe "rt_host_alloc"      c_host_alloc      :: CSize -> IO (Ptr Word8)
foreign import ccall        "&rt_host_free"      p_host_free       :: FunPtr (Ptr Word8 -> IO ())
-- long calls that may call back into Haskell (fnorm): `safe`, so other Haskell threads keep running and the callback may enter
foreign import ccall safe   "rt_render"          c_render          :: ScenePtr -> CInt -> CInt -> CInt -> CInt -> Word64 -> CInt -> CInt -> Ptr Word8 -> Ptr Int32 -> IO CInt
foreign import ccall safe   "rt_render_multi"    c_render_multi    :: ScenePtr -> CInt -> CInt -> CInt -> CInt -> Word64 -> CInt -> Ptr CInt -> Ptr Word8 -> Ptr Int32 -> IO CInt

check :: String -> CInt -> IO CInt
check what rc = do
  when (rc < 0) $ do
    msg <- c_last_error >>= peekCString
    ioError (userError (what ++ ": " ++ msg))
  return rc

-- | rt_init: the HIP device this OS thread renders on (optional; device 0 otherwise).  Call it from a bound thread.
initDevice :: Int -> IO ()
initDevice d = void $ c_init (fromIntegral d) >>= check "rt_init"

-- rt_material: diffuse[3], specular[3], 2 bytes of padding, specularity, reflectivity (24 bytes, 8-aligned)
withMaterials :: [Material] -> (Ptr Word8 -> IO a) -> IO a
withMaterials ms act = allocaBytesAligned (24 * max 1 (length ms)) 8 $ \p -> do
  forM_ (zip [0 ..] ms) $ \(i, MkMaterial (PixelRGB8 dr dg db) (PixelRGB8 sr sg sb) sp rf) -> do
    let q = p `plusPtr` (24 * i)
    pokeByteOff q 0 dr >> pokeByteOff q 1 dg >> pokeByteOff q 2 db
    pokeByteOff q 3 sr >> pokeByteOff q 4 sg >> pokeByteOff q 5 sb
    pokeByteOff q 6 (0 :: Word8) >> pokeByteOff q 7 (0 :: Word8)
    pokeByteOff q 8 (realToFrac sp :: CDouble)
    pokeByteOff q 16 (realToFrac rf :: CDouble)
  act p

withV3 :: Vector3D -> (Ptr CDouble -> IO a) -> IO a
withV3 (MkV3D x y z) = withArray (map realToFrac [x, y, z])

-- | Replays the recorded tree in depth-first order: the leaf order of the library (and the value in the hit-ID buffer)
-- is the reference's tie order (left-most leaf wins equal t, Main.hs:213-214).  FunPtrs made for fnorm closures are
-- collected in `keep` and freed when the scene is destroyed.
emit :: IORef [FunPtr NormalFn] -> ScenePtr -> Shape -> IO ()
emit keep s shape = case shape of
  Append a b -> do
    void $ c_begin_mappend s >>= check "rt_begin_mappend"
    emit keep s a >> emit keep s b
    void $ c_end_mappend s >>= check "rt_end_mappend"
  Concat xs -> do
    void $ c_begin_mconcat s >>= check "rt_begin_mconcat"
    mapM_ (emit keep s) xs
    void $ c_end_mconcat s >>= check "rt_end_mconcat"
  Rect m p w h -> withMaterials [m] $ \pm -> withV3 p $ \pp -> withV3 w $ \pw -> withV3 h $ \ph ->
    void $ c_add_rectangle s pm pp pw ph >>= check "rt_add_rectangle"
  Tri m a b c -> withMaterials [m] $ \pm -> withV3 a $ \pa -> withV3 b $ \pb -> withV3 c $ \pc ->
    void $ c_add_triangle s pm pa pb pc >>= check "rt_add_triangle"
  Sph m c r -> withMaterials [m] $ \pm -> withV3 c $ \pc ->
    void $ c_add_sphere s pm pc (realToFrac r) >>= check "rt_add_sphere"
  Cuboid ms p l h d -> withMaterials ms $ \pm -> withV3 p $ \pp ->
    void $ c_add_cuboid s pm (fromIntegral (length ms)) pp (realToFrac l) (realToFrac h) (realToFrac d) >>= check "rt_add_cuboid"
  Cyl t b m p n h r -> withMaterials [t] $ \pt -> withMaterials [b] $ \pb -> withMaterials [m] $ \pm -> withV3 p $ \pp ->
    void $ c_add_cylinder s pt pb pm pp (fromIntegral n) (realToFrac h) (realToFrac r) >>= check "rt_add_cylinder"
  Normals fnorm' scaleInv (MkV3D ox oy oz) child -> do
    first <- c_num_leaves s
    emit keep s child
    n <- c_num_leaves s
    -- the library hands over the arguments (x, y) of fnorm for the hit points it found (Main.hs:961) and takes n' back
    fp <- mkNormalFn $ \cnt xy out _ -> forM_ [0 .. fromIntegral cnt - 1] $ \i -> do
      x <- peekElemOff xy (2 * i)
      y <- peekElemOff xy (2 * i + 1)
      let MkV3D nx ny nz = fnorm' (realToFrac x) (realToFrac y)
      pokeElemOff out (3 * i) (realToFrac nx) >> pokeElemOff out (3 * i + 1) (realToFrac ny) >> pokeElemOff out (3 * i + 2) (realToFrac nz)
    modifyIORef keep (fp :)
    withArray (map realToFrac [scaleInv, ox, oy, oz]) $ \pp ->
      void $ c_set_normal_callback s first (n - first) pp fp nullPtr >>= check "rt_set_normal_callback"

-- The lights.  "The same world" is decided by identity of the heap object (StableName), which is what `pointLight world`
-- with the `world` that is then traced gives; a Shape that merely looks the same becomes a shadow world (same picture,
-- twice the leaves).  Shadow worlds are emitted once per distinct Shape, after the traced world, before the lights.
emitLights :: IORef [FunPtr NormalFn] -> ScenePtr -> Shape -> [LightSrc] -> IO ()
emitLights keep s world ls = do
  traced <- makeStableName $! world                       -- ($!: an unevaluated thunk and its value may get different names)
  known  <- newIORef []                                   -- [(StableName Shape, world number)]
  worlds <- forM ls $ \l -> case l of
    Ambient _ -> return 0
    Point lw _ _ _ -> do
      sn <- makeStableName $! lw
      if sn == traced then return 0 else do
        seen <- readIORef known
        case lookup sn seen of
          Just k  -> return k
          Nothing -> do
            k <- c_begin_shadow_world s >>= check "rt_begin_shadow_world"
            emit keep s lw
            void $ c_end_shadow_world s >>= check "rt_end_shadow_world"
            modifyIORef known ((sn, k) :)
            return k
  forM_ (zip ls worlds) $ \(l, k) -> case l of
    Ambient f -> void $ c_add_ambient s (realToFrac f) >>= check "rt_add_ambient"
    Point _ kd ks p -> withV3 p $ \pp -> void $
      (if k == 0 then c_add_point_light s (realToFrac kd) (realToFrac ks) pp
                 else c_add_point_light_world s k (realToFrac kd) (realToFrac ks) pp) >>= check "rt_add_point_light"

withScene :: Light -> Shape -> (ScenePtr -> IO a) -> IO a
withScene (MkLight ls) world act = do
  keep <- newIORef []
  bracket (alloca $ \pp -> c_scene_create pp >>= check "rt_scene_create" >> peek pp)
          (\s -> c_scene_destroy s >> readIORef keep >>= mapM_ freeHaskellFunPtr) $ \s -> do
    emit keep s world
    emitLights keep s world ls
    act s

hostBuffer :: Int -> IO (ForeignPtr Word8)
hostBuffer bytes = do
  -- page-locked memory from the library (freed by rt_host_free when the ForeignPtr dies): the copy back is DMA behind
  -- the render; mallocForeignPtrBytes would work too, through the runtime's bounce buffers
  raw <- c_host_alloc (fromIntegral (max 1 bytes))
  when (raw == nullPtr) $ c_last_error >>= peekCString >>= \m -> ioError (userError ("rt_host_alloc: " ++ m))
  newForeignPtr p_host_free raw

-- | Rows [row0, row0 + nrows) of the w x h image as RGBA8 (row-major, y = 0 is the top row): what one rank of a
-- one-process-per-GPU job renders.
renderBand :: Int -> Int -> Int -> Light -> Shape -> Int -> Int -> IO (ForeignPtr Word8)
renderBand w h depth lights world row0 nrows = withScene lights world $ \s -> do
  -- RT_OPT_RGB24 (= 7): 3 bytes per pixel, the PixelRGB8 generateImage wants; a quarter less over PCIe than RGBA8
  void $ c_scene_set_option s 7 1 >>= check "rt_scene_set_option"
  buf <- hostBuffer (w * nrows * 3)
  withForeignPtr buf $ \p ->
    void $ c_render s (fromIntegral w) (fromIntegral h) (fromIntegral depth) 1 0 (fromIntegral row0) (fromIntegral nrows) p nullPtr >>= check "rt_render"
  return buf

pixelAt :: ForeignPtr Word8 -> Int -> Int -> Int -> PixelRGB8
pixelAt buf w x y = unsafePerformIO $ withForeignPtr buf $ \p ->
  PixelRGB8 <$> peekByteOff p (3*(y*w+x)) <*> peekByteOff p (3*(y*w+x)+1) <*> peekByteOff p (3*(y*w+x)+2)

-- | Main.hs:623-637.  `parallelTrace w h camera (trace n lights world)` renders the frame once, on the GPU, when the
-- first pixel is demanded (the reference forces its Repa array at that moment), and serves every (x, y) from it.
parallelTrace :: Int -> Int -> Camera -> Tracer -> Int -> Int -> PixelRGB8
parallelTrace w h _camera (MkTracer depth lights world) =
  let frame = unsafePerformIO (renderBand w h depth lights world 0 h)
  in \x y -> pixelAt frame w x y
{-# NOINLINE parallelTrace #-}

-- | The same frame on several GPUs of this process at once (rt_render_multi): contiguous row bands of equal measured
-- cost, one per device, gathered on the first device over xGMI and copied to host memory from there.
parallelTraceMulti :: [Int] -> Int -> Int -> Camera -> Tracer -> Int -> Int -> PixelRGB8
parallelTraceMulti devices w h _camera (MkTracer depth lights world) =
  let frame = unsafePerformIO $ withScene lights world $ \s -> do
        void $ c_scene_set_option s 7 1 >>= check "rt_scene_set_option"      -- RT_OPT_RGB24
        buf <- hostBuffer (w * h * 3)
        withForeignPtr buf $ \p -> withArray (map fromIntegral devices) $ \pd ->
          void $ c_render_multi s (fromIntegral w) (fromIntegral h) (fromIntegral depth) 1 0 (fromIntegral (length devices)) pd p nullPtr >>= check "rt_render_multi"
        return buf
  in \x y -> pixelAt frame w x y
{-# NOINLINE parallelTraceMulti #-}
