## Drop-in replacements for the hot functions of the reference's Computing_functions.R / MAGMA.R,
## calling libmagma_hip.so through the .Call shim in r_shim.c.
## NOT exercised in this repository's tests: R is not installed in the build image. The tested
## host is the Python mirror in magma_amd/api.py, which drives the same C ABI.
##
## Usage: source the reference's scripts first, then this file (it overrides e_step,
## logL_GP_mod, gr_GP_mod, logL_GP_mod_common_hp, gr_GP_mod_common_hp, logL_monitoring,
## posterior_mu, and m_step for individual hyper-parameters); `training`, `full_algo`, `pred_gp`,
## `train_new_gp` keep the reference's bodies and pick these up (opm keeps calling fn / gr per
## evaluation for theta_0 and for common hyper-parameters).
## Only the SE kernel (`kernel` / `kernel_mu`) is accelerated; any other kernel falls back to
## the reference's R code, kept under the names *_ref.

## The native driver (one .Call per reference function) is used for `training`, `m_step`,
## `train_new_gp`; set options(magmahip.native = FALSE) to keep R's loops over the per-evaluation
## calls (e.g. to let `opm` use another optimiser).
dyn.load(file.path(Sys.getenv("MAGMAHIP_DIR", "."), "magmahip.so"))
.hip_native <- function() isTRUE(getOption("magmahip.native", TRUE))

## db -> native dataset handle (rows of an individual contiguous, Timestamp ascending)
.hip_dataset <- function(db, extra_t = NULL) {
  db$ID <- as.character(db$ID)
  ids <- unique(db$ID)
  rows <- lapply(ids, function(i) { d <- db[db$ID == i, ]; d[order(d$Timestamp), ] })
  n <- vapply(rows, nrow, 1L)
  t <- unlist(lapply(rows, function(d) d$Timestamp))
  y <- unlist(lapply(rows, function(d) d$Output))
  grid <- sort(unique(c(t, extra_t)))
  list(ptr = .Call("magmahip_dataset", as.integer(c(0L, cumsum(n))), as.double(t), as.double(y),
                   as.double(grid)), ids = ids, grid = grid, off = as.integer(c(0L, cumsum(n))), t = t, y = y)
}

.hip_is_se <- function(k) identical(k, kernel) || identical(k, kernel_mu)

## db -> session: ragged layout (per-ID rows sorted by Timestamp) + sorted union grid + index
.hip_session <- function(db, extra_t = NULL) {
  db$ID <- as.character(db$ID)
  ids <- unique(db$ID)
  rows <- lapply(ids, function(i) {
    d <- db[db$ID == i, ]
    d[order(d$Timestamp), ]
  })
  n <- vapply(rows, nrow, 1L)
  if (max(n) > 64) stop("magmahip: at most 64 observations per individual")
  t <- unlist(lapply(rows, function(d) d$Timestamp))
  y <- unlist(lapply(rows, function(d) d$Output))
  grid <- sort(unique(c(t, extra_t)))
  list(ptr = .Call("magmahip_session", as.integer(c(0L, cumsum(n))), as.double(t), as.double(y),
                   as.integer(match(t, grid) - 1L), as.double(grid)),
       ids = ids, grid = grid, n = n, key = digest_key(db))
}
digest_key <- function(db) paste(nrow(db), sum(db$Timestamp), sum(db$Output))
.hip_cache <- new.env()
.hip_get <- function(db, extra_t = NULL) {
  key <- paste(digest_key(db), length(extra_t), sum(extra_t))
  if (is.null(.hip_cache[[key]])) assign(key, .hip_session(db, extra_t), envir = .hip_cache)
  .hip_cache[[key]]
}
.hip_theta_mat <- function(theta_i, ids) sapply(ids, function(i) as.numeric(unlist(theta_i[[i]]))[1:3])

e_step_ref <- e_step
e_step <- function(db, m_0, kern_0, kern_i, hp) {
  if (!(.hip_is_se(kern_0) && .hip_is_se(kern_i))) return(e_step_ref(db, m_0, kern_0, kern_i, hp))
  s <- .hip_get(db)
  th <- .hip_theta_mat(hp$theta_i, s$ids)
  pen <- mean(th[3, ])                                           # CF:381
  r <- .Call("magmahip_e_step", s$ptr, as.numeric(unlist(hp$theta_0))[1:2], pen, th, as.double(m_0))
  if (r[[3]] != 0) return(e_step_ref(db, m_0, kern_0, kern_i, hp)) # solve()/ginv() slow path
  nm <- paste0("X", s$grid)
  cov <- r[[2]]; dimnames(cov) <- list(nm, nm)
  list(mean = tibble::tibble(Timestamp = s$grid, Output = r[[1]]), cov = cov,
       pred_GP = tibble::tibble(Timestamp = s$grid, Mean = r[[1]], Var = diag(cov)))
}

posterior_mu_ref <- posterior_mu
posterior_mu <- function(db, new_db, timestamps, m_0, kern_0, kern_i, hp, pen_diag = 0.1) {
  if (!(.hip_is_se(kern_0) && .hip_is_se(kern_i))) return(posterior_mu_ref(db, new_db, timestamps, m_0, kern_0, kern_i, hp))
  s <- .hip_get(db, c(timestamps, new_db$Timestamp))
  th <- .hip_theta_mat(hp$theta_i, s$ids)
  r <- .Call("magmahip_e_step", s$ptr, as.numeric(unlist(hp$theta_0))[1:2], pen_diag, th, as.double(m_0))
  if (r[[3]] != 0) return(posterior_mu_ref(db, new_db, timestamps, m_0, kern_0, kern_i, hp))
  nm <- paste0("X", s$grid)
  cov <- r[[2]]; dimnames(cov) <- list(nm, nm)
  list(mean = tibble::tibble(Timestamp = s$grid, Output = r[[1]]), cov = cov,
       pred_GP = tibble::tibble(Timestamp = s$grid, Mean = r[[1]], Var = diag(cov)))
}

## Sum over individuals with a common triple (CF:203-234, 334-368)
.hip_common <- function(hp, db, mean, new_cov, want_grad) {
  s <- .hip_get(db, mean$Timestamp)
  .Call("magmahip_set_posterior", s$ptr, mean$Output[order(mean$Timestamp)], new_cov)
  r <- .Call("magmahip_indiv_eval", s$ptr, as.numeric(hp)[1:3], want_grad, 0L)
  if (any(r[[3]] != 0)) return(NULL)
  if (want_grad) rowSums(r[[2]]) else sum(r[[1]])
}
logL_GP_mod_common_hp_ref <- logL_GP_mod_common_hp
logL_GP_mod_common_hp <- function(hp, db, mean, kern, new_cov) {
  v <- if (.hip_is_se(kern)) .hip_common(hp, db, mean, new_cov, FALSE) else NULL
  if (is.null(v)) logL_GP_mod_common_hp_ref(hp, db, mean, kern, new_cov) else v
}
gr_GP_mod_common_hp_ref <- gr_GP_mod_common_hp
gr_GP_mod_common_hp <- function(hp, db, mean, kern, new_cov) {
  v <- if (.hip_is_se(kern)) .hip_common(hp, db, mean, new_cov, TRUE) else NULL
  if (is.null(v)) gr_GP_mod_common_hp_ref(hp, db, mean, kern, new_cov) else v
}

## theta_0 objective (length(hp) == 2): db = posterior mean tibble, new_cov = N x N
logL_GP_mod_ref <- logL_GP_mod
logL_GP_mod <- function(hp, db, mean, kern, new_cov, pen_diag = NULL) {
  if (length(hp) != 2 || !.hip_is_se(kern)) return(logL_GP_mod_ref(hp, db, mean, kern, new_cov, pen_diag))
  s <- .hip_get(data.frame(ID = character(0), Timestamp = numeric(0), Output = numeric(0)), db$Timestamp)
  .Call("magmahip_set_posterior", s$ptr, db$Output[order(db$Timestamp)], new_cov)
  r <- .Call("magmahip_theta0_eval", s$ptr, as.numeric(hp), pen_diag, as.double(mean), FALSE)
  if (r[4] != 0) logL_GP_mod_ref(hp, db, mean, kern, new_cov, pen_diag) else r[1]
}
gr_GP_mod_ref <- gr_GP_mod
gr_GP_mod <- function(hp, db, mean, kern, new_cov, pen_diag = NULL) {
  if (length(hp) != 2 || !.hip_is_se(kern)) return(gr_GP_mod_ref(hp, db, mean, kern, new_cov, pen_diag))
  s <- .hip_get(data.frame(ID = character(0), Timestamp = numeric(0), Output = numeric(0)), db$Timestamp)
  .Call("magmahip_set_posterior", s$ptr, db$Output[order(db$Timestamp)], new_cov)
  r <- .Call("magmahip_theta0_eval", s$ptr, as.numeric(hp), pen_diag, as.double(mean), TRUE)
  if (r[4] != 0) gr_GP_mod_ref(hp, db, mean, kern, new_cov, pen_diag) else r[2:3]
}

logL_monitoring_ref <- logL_monitoring
logL_monitoring <- function(hp, db, kern_i, kern_0, mean_mu, cov_mu, m_0) {
  if (!(.hip_is_se(kern_0) && .hip_is_se(kern_i))) return(logL_monitoring_ref(hp, db, kern_i, kern_0, mean_mu, cov_mu, m_0))
  s <- .hip_get(db, mean_mu$Timestamp)
  th <- .hip_theta_mat(hp$theta_i, s$ids)
  .Call("magmahip_set_posterior", s$ptr, mean_mu$Output[order(mean_mu$Timestamp)], cov_mu)
  r0 <- .Call("magmahip_theta0_eval", s$ptr, as.numeric(unlist(hp$theta_0))[1:2], mean(th[3, ]), as.double(m_0), FALSE)
  ri <- .Call("magmahip_indiv_eval", s$ptr, th, FALSE, 0L)
  if (r0[4] != 0 || any(ri[[3]] != 0)) return(logL_monitoring_ref(hp, db, kern_i, kern_0, mean_mu, cov_mu, m_0))
  -r0[1] - sum(ri[[1]])
}

## m_step (CF:397-437) with individual hyper-parameters: the M independent optim() calls of
## CF:423-431 become ONE .Call (magma_indiv_mstep: batched lock-step L-BFGS-B in the library);
## theta_0 keeps going through opm with the GPU fn / gr above. common_hp = TRUE and non-SE
## kernels keep the reference's body.
m_step_ref <- m_step
m_step <- function(db, old_hp, mean, cov, kern_0, kern_i, m_0, common_hp) {
  if (!(.hip_is_se(kern_0) && .hip_is_se(kern_i)))
    return(m_step_ref(db, old_hp, mean, cov, kern_0, kern_i, m_0, common_hp))
  if (.hip_native()) {   # both branches in ONE call: magma_m_step
    s <- .hip_dataset(db, mean$Timestamp)
    th <- .hip_theta_mat(old_hp$theta_i, s$ids)
    r <- .Call("magmahip_m_step", s$ptr, mean$Output[order(mean$Timestamp)], cov, as.double(m_0),
               as.numeric(unlist(old_hp$theta_0))[1:2], th, common_hp)
    if (anyNA(r[[1]]) || anyNA(r[[2]])) return(m_step_ref(db, old_hp, mean, cov, kern_0, kern_i, m_0, common_hp))
    new_theta_i <- lapply(seq_along(s$ids), function(j) setNames(r[[2]][, j], names(old_hp$theta_i[[1]])))
    names(new_theta_i) <- s$ids
    return(list(theta_0 = setNames(r[[1]], names(old_hp$theta_0)), theta_i = new_theta_i))
  }
  if (common_hp) return(m_step_ref(db, old_hp, mean, cov, kern_0, kern_i, m_0, common_hp))
  s <- .hip_get(db, mean$Timestamp)
  th <- .hip_theta_mat(old_hp$theta_i, s$ids)
  pen_diag <- mean(2 * th[3, ])                                              # CF:409
  new_theta_0 <- opm(old_hp$theta_0, logL_GP_mod, gr_GP_mod, db = mean, mean = m_0, kern = kern_0,
                     new_cov = cov, pen_diag = pen_diag, method = "L-BFGS-B",
                     control = list(kkt = FALSE))[1, 1:2]
  .Call("magmahip_set_posterior", s$ptr, mean$Output[order(mean$Timestamp)], cov)
  r <- .Call("magmahip_indiv_mstep", s$ptr, th, 0L)
  if (any(r[[4]] == 5)) return(m_step_ref(db, old_hp, mean, cov, kern_0, kern_i, m_0, common_hp))
  new_theta_i <- lapply(seq_along(s$ids), function(j) setNames(r[[1]][, j], names(old_hp$theta_i[[1]])))
  names(new_theta_i) <- s$ids
  list(theta_0 = new_theta_0, theta_i = new_theta_i)
}

## training (MAGMA.R:15-82) as ONE .Call: the EM loop, both optimisers, the monitoring and the
## stopping rule run inside the library (magma_training). Same return value as the reference.
training_ref <- training
training <- function(db, prior_mean, ini_hp, kern_0, kern_i, common_hp = T) {
  if (!.hip_native() || !(.hip_is_se(kern_0) && .hip_is_se(kern_i)))
    return(training_ref(db, prior_mean, ini_hp, kern_0, kern_i, common_hp))
  t1 <- Sys.time()
  s <- .hip_dataset(db)
  r <- .Call("magmahip_training", s$ptr, as.double(prior_mean), as.numeric(unlist(ini_hp$theta_0))[1:2],
             as.numeric(unlist(ini_hp$theta_i))[1:3], common_hp)
  theta_i <- lapply(seq_along(s$ids), function(j) r[[2]][, j])
  names(theta_i) <- s$ids
  nm <- paste0("X", s$grid)
  cov <- r[[7]]; dimnames(cov) <- list(nm, nm)
  param <- list(mean = tibble::tibble(Timestamp = s$grid, Output = r[[6]]), cov = cov,
                pred_GP = tibble::tibble(Timestamp = s$grid, Mean = r[[6]], Var = diag(cov)))
  list(hp = list(theta_0 = r[[1]], theta_i = theta_i), convergence = r[[3]], param = param,
       Time_train = Sys.time() - t1, plot = list(), logLL_trace = r[[5]][, seq_len(r[[4]]), drop = FALSE])
}

## train_new_gp (MAGMA.R:84-104): the 3-parameter L-BFGS-B inside the library (n <= 64)
train_new_gp_ref <- train_new_gp
train_new_gp <- function(db, mean_mu, cov_mu, ini_hp_i, kern_i) {
  if (!.hip_native() || !.hip_is_se(kern_i) || nrow(db) > 64 || is.null(dim(cov_mu)))
    return(train_new_gp_ref(db, mean_mu, cov_mu, ini_hp_i, kern_i))
  db <- db[order(db$Timestamp), ]
  mean <- if (is.vector(mean_mu)) rep(mean_mu, length.out = nrow(db)) else
    mean_mu$Output[match(db$Timestamp, mean_mu$Timestamp)]
  nm <- paste0("X", db$Timestamp)
  r <- .Call("magmahip_train_new_gp", as.double(db$Timestamp), as.double(db$Output), as.double(mean),
             cov_mu[nm, nm], as.numeric(unlist(ini_hp_i))[1:3])
  list(theta = r[1:2], sigma = r[3])
}
